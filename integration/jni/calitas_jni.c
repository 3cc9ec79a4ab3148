/* calitas_jni.c -- JNI glue between com.editasmedicine.aligner.NativeAligner (integration/scala/NativeAligner.scala) and the
 * C ABI of include/calitas_hip.h.  NOT compiled in this repository (no JDK / jni.h in the build image); build on a box
 * with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include calitas_jni.c \
 *       -L../../calitas_amd -lcalitas_hip -Wl,-rpath,'$ORIGIN' -o libcalitas_jni.so
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "calitas_hip.h"

static void throw_state(JNIEnv* env, const char* msg) {
  (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalStateException"), msg ? msg : "calitas error");
}

JNIEXPORT jlong JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_create(JNIEnv* env, jobject self, jint device) {
  calitas_ctx* ctx = NULL;
  if (calitas_create(device, &ctx) != CALITAS_OK) { throw_state(env, calitas_last_error(NULL)); return 0; }
  return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_destroy(JNIEnv* env, jobject self, jlong h) {
  calitas_destroy((calitas_ctx*)(intptr_t)h);
}

/* names: String[]; bases: byte[][] as read by ReferenceSequenceIterator (SearchReference.scala:41-49). */
JNIEXPORT void JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_setReference(JNIEnv* env, jobject self, jlong h,
    jobjectArray names, jobjectArray bases, jstring genomeBuild) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  const jsize n = (*env)->GetArrayLength(env, names);
  const char** cnames = calloc((size_t)n, sizeof(char*));
  uint64_t* lens = calloc((size_t)n, sizeof(uint64_t));
  const uint8_t** ptrs = calloc((size_t)n, sizeof(uint8_t*));
  jbyteArray* arrs = calloc((size_t)n, sizeof(jbyteArray));
  jstring* jn = calloc((size_t)n, sizeof(jstring));
  const char* build = genomeBuild ? (*env)->GetStringUTFChars(env, genomeBuild, NULL) : "unknown";
  for (jsize i = 0; i < n; i++) {
    jn[i] = (jstring)(*env)->GetObjectArrayElement(env, names, i);
    cnames[i] = (*env)->GetStringUTFChars(env, jn[i], NULL);
    arrs[i] = (jbyteArray)(*env)->GetObjectArrayElement(env, bases, i);
    lens[i] = (uint64_t)(*env)->GetArrayLength(env, arrs[i]);
    /* inputs are borrowed only for the duration of calitas_set_reference, which keeps its own 2-bit copy */
    ptrs[i] = (const uint8_t*)(*env)->GetByteArrayElements(env, arrs[i], NULL);
  }
  const int rc = calitas_set_reference(ctx, (int32_t)n, cnames, lens, ptrs, build);
  for (jsize i = 0; i < n; i++) {
    (*env)->ReleaseByteArrayElements(env, arrs[i], (jbyte*)ptrs[i], JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, jn[i], cnames[i]);
  }
  if (genomeBuild) (*env)->ReleaseStringUTFChars(env, genomeBuild, build);
  free(cnames); free(lens); free(ptrs); free(arrs); free(jn);
  if (rc != CALITAS_OK) throw_state(env, calitas_last_error(ctx));
}

/* Round 5: the same with the contigs' lengths given and null for the contigs this process holds no bases of (calitas_set_reference with
 * absent contigs: what one process of a multi-GPU job does with the contigs its window range does not touch). */
JNIEXPORT void JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_setReferenceWithLengths(JNIEnv* env, jobject self, jlong h,
    jobjectArray names, jlongArray lengths, jobjectArray bases, jstring genomeBuild) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  const jsize n = (*env)->GetArrayLength(env, names);
  if ((*env)->GetArrayLength(env, lengths) != n || (*env)->GetArrayLength(env, bases) != n) { throw_state(env, "names, lengths and bases differ in length"); return; }
  const char** cnames = calloc((size_t)n, sizeof(char*));
  uint64_t* lens = calloc((size_t)n, sizeof(uint64_t));
  const uint8_t** ptrs = calloc((size_t)n, sizeof(uint8_t*));
  jbyteArray* arrs = calloc((size_t)n, sizeof(jbyteArray));
  jstring* jn = calloc((size_t)n, sizeof(jstring));
  jlong* jl = (*env)->GetLongArrayElements(env, lengths, NULL);
  const char* build = genomeBuild ? (*env)->GetStringUTFChars(env, genomeBuild, NULL) : "unknown";
  int bad = 0;
  for (jsize i = 0; i < n; i++) {
    jn[i] = (jstring)(*env)->GetObjectArrayElement(env, names, i);
    cnames[i] = (*env)->GetStringUTFChars(env, jn[i], NULL);
    lens[i] = (uint64_t)jl[i];
    arrs[i] = (jbyteArray)(*env)->GetObjectArrayElement(env, bases, i);
    if (arrs[i]) {
      if ((uint64_t)(*env)->GetArrayLength(env, arrs[i]) != lens[i]) bad = 1;
      ptrs[i] = (const uint8_t*)(*env)->GetByteArrayElements(env, arrs[i], NULL);
    }
  }
  const int rc = bad ? CALITAS_EINVAL : calitas_set_reference(ctx, (int32_t)n, cnames, lens, ptrs, build);
  for (jsize i = 0; i < n; i++) {
    if (arrs[i]) (*env)->ReleaseByteArrayElements(env, arrs[i], (jbyte*)ptrs[i], JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, jn[i], cnames[i]);
  }
  (*env)->ReleaseLongArrayElements(env, lengths, jl, JNI_ABORT);
  if (genomeBuild) (*env)->ReleaseStringUTFChars(env, genomeBuild, build);
  free(cnames); free(lens); free(ptrs); free(arrs); free(jn);
  if (bad) throw_state(env, "a contig's bases are not as long as its length says");
  else if (rc != CALITAS_OK) throw_state(env, calitas_last_error(ctx));
}

/* Fills calitas_params_t from the caller's int[]: the fields in declaration order.  The struct has grown over time (13 ints at first,
 * then first_window / n_windows); an older caller's shorter array leaves the trailing fields at zero, which is their "whole job"
 * default.  Fewer than the original 13 is a caller bug.  Returns 0 on failure with a Java exception pending. */
#define CALITAS_PARAMS_MIN_INTS 13
static int read_params(JNIEnv* env, jintArray params, calitas_params_t* p) {
  memset(p, 0, sizeof(*p));
  if (!params) { throw_state(env, "params is null"); return 0; }
  const jsize have = (*env)->GetArrayLength(env, params);
  const jsize want = (jsize)(sizeof(*p) / sizeof(int32_t));
  if (have < CALITAS_PARAMS_MIN_INTS) { throw_state(env, "params: at least 13 ints (the fields of calitas_params_t in declaration order)"); return 0; }
  (*env)->GetIntArrayRegion(env, params, 0, have < want ? have : want, (jint*)p);
  return (*env)->ExceptionCheck(env) ? 0 : 1;
}

/* params: the int fields of calitas_params_t in declaration order (read_params above).  Returns a direct ByteBuffer over the
 * library-owned calitas_aln_t array; NativeAligner hands the address back to `free` when done. */
JNIEXPORT jobject JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_search(JNIEnv* env, jobject self, jlong h,
    jstring protospacer, jobjectArray pams, jboolean pam5, jint cliLength, jintArray params) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  calitas_guide_t g;
  calitas_params_t p;
  if (!read_params(env, params, &p)) return NULL;          /* before anything is acquired: nothing to release on failure */
  const jsize np = (*env)->GetArrayLength(env, pams);
  const char* cp[CALITAS_MAX_PAMS];
  jstring jp[CALITAS_MAX_PAMS];
  if (np > CALITAS_MAX_PAMS) { throw_state(env, "too many PAMs"); return NULL; }
  g.protospacer = (*env)->GetStringUTFChars(env, protospacer, NULL);
  for (jsize i = 0; i < np; i++) { jp[i] = (jstring)(*env)->GetObjectArrayElement(env, pams, i); cp[i] = (*env)->GetStringUTFChars(env, jp[i], NULL); }
  g.n_pams = (int32_t)np; g.pams = cp; g.pam_is_5prime = pam5 ? 1 : 0; g.cli_length = cliLength;
  calitas_aln_t* alns = NULL;
  uint64_t n = 0;
  const int rc = calitas_search(ctx, 1, &g, &p, &alns, &n);
  (*env)->ReleaseStringUTFChars(env, protospacer, g.protospacer);
  for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, jp[i], cp[i]);
  if (rc != CALITAS_OK) { throw_state(env, calitas_last_error(ctx)); return NULL; }
  return (*env)->NewDirectByteBuffer(env, alns, (jlong)(n * sizeof(calitas_aln_t)));
}

/* calitas_search_hits: the finished hits.txt text (header + rows) for one guide as a direct ByteBuffer over the library-owned
 * (page-locked) text; `free` releases it.  version may be null (then "unknown-<date>" like EditasMetric.Version without a jar). */
JNIEXPORT jobject JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_searchHits(JNIEnv* env, jobject self, jlong h,
    jstring protospacer, jobjectArray pams, jboolean pam5, jint cliLength, jstring guideId, jintArray params, jstring version) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  calitas_guide_t g;
  calitas_params_t p;
  if (!read_params(env, params, &p)) return NULL;          /* before anything is acquired: nothing to release on failure */
  const jsize np = (*env)->GetArrayLength(env, pams);
  const char* cp[CALITAS_MAX_PAMS];
  jstring jp[CALITAS_MAX_PAMS];
  if (np > CALITAS_MAX_PAMS) { throw_state(env, "too many PAMs"); return NULL; }
  g.protospacer = (*env)->GetStringUTFChars(env, protospacer, NULL);
  for (jsize i = 0; i < np; i++) { jp[i] = (jstring)(*env)->GetObjectArrayElement(env, pams, i); cp[i] = (*env)->GetStringUTFChars(env, jp[i], NULL); }
  g.n_pams = (int32_t)np; g.pams = cp; g.pam_is_5prime = pam5 ? 1 : 0; g.cli_length = cliLength;
  const char* gid = (*env)->GetStringUTFChars(env, guideId, NULL);
  const char* ver = version ? (*env)->GetStringUTFChars(env, version, NULL) : NULL;
  char* tsv = NULL;
  uint64_t bytes = 0, rows = 0;
  const int rc = calitas_search_hits(ctx, &g, gid, &p, ver, NULL, &tsv, &bytes, &rows);
  (*env)->ReleaseStringUTFChars(env, protospacer, g.protospacer);
  for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, jp[i], cp[i]);
  (*env)->ReleaseStringUTFChars(env, guideId, gid);
  if (version) (*env)->ReleaseStringUTFChars(env, version, ver);
  if (rc != CALITAS_OK) { throw_state(env, calitas_last_error(ctx)); return NULL; }
  return (*env)->NewDirectByteBuffer(env, tsv, (jlong)bytes);
}

/* calitas_search_variants: SearchReference.execute with --variants (SearchReference.scala:570-648); vcfId is the "name:md5" string
 * of ReferenceHit.scala:175-183, chrom the --chrom filter or null.  Returns the finished hits.txt text like searchHits. */
JNIEXPORT jobject JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_searchVariants(JNIEnv* env, jobject self, jlong h,
    jstring protospacer, jobjectArray pams, jboolean pam5, jint cliLength, jstring guideId, jintArray params, jstring vcfPath,
    jstring chrom, jstring vcfId, jstring version) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  calitas_guide_t g;
  calitas_params_t p;
  if (!read_params(env, params, &p)) return NULL;          /* before anything is acquired: nothing to release on failure */
  const jsize np = (*env)->GetArrayLength(env, pams);
  const char* cp[CALITAS_MAX_PAMS];
  jstring jp[CALITAS_MAX_PAMS];
  if (np > CALITAS_MAX_PAMS) { throw_state(env, "too many PAMs"); return NULL; }
  g.protospacer = (*env)->GetStringUTFChars(env, protospacer, NULL);
  for (jsize i = 0; i < np; i++) { jp[i] = (jstring)(*env)->GetObjectArrayElement(env, pams, i); cp[i] = (*env)->GetStringUTFChars(env, jp[i], NULL); }
  g.n_pams = (int32_t)np; g.pams = cp; g.pam_is_5prime = pam5 ? 1 : 0; g.cli_length = cliLength;
  const char* gid = (*env)->GetStringUTFChars(env, guideId, NULL);
  const char* vcf = (*env)->GetStringUTFChars(env, vcfPath, NULL);
  const char* vid = (*env)->GetStringUTFChars(env, vcfId, NULL);
  const char* chr = chrom ? (*env)->GetStringUTFChars(env, chrom, NULL) : NULL;
  const char* ver = version ? (*env)->GetStringUTFChars(env, version, NULL) : NULL;
  char* tsv = NULL;
  uint64_t bytes = 0, rows = 0;
  const int rc = calitas_search_variants(ctx, &g, gid, &p, vcf, chr, vid, ver, NULL, &tsv, &bytes, &rows, NULL);
  (*env)->ReleaseStringUTFChars(env, protospacer, g.protospacer);
  for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, jp[i], cp[i]);
  (*env)->ReleaseStringUTFChars(env, guideId, gid);
  (*env)->ReleaseStringUTFChars(env, vcfPath, vcf);
  (*env)->ReleaseStringUTFChars(env, vcfId, vid);
  if (chrom) (*env)->ReleaseStringUTFChars(env, chrom, chr);
  if (version) (*env)->ReleaseStringUTFChars(env, version, ver);
  if (rc != CALITAS_OK) { throw_state(env, calitas_last_error(ctx)); return NULL; }
  return (*env)->NewDirectByteBuffer(env, tsv, (jlong)bytes);
}

/* Round 5.  calitas_alloc_host: a page-locked block of the runtime's own as a direct ByteBuffer -- the destination searchVariantsInto
 * likes best (the GPU's copy engines write into it directly); kept by the caller for the life of the process, released with `free`. */
JNIEXPORT jobject JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_allocHost(JNIEnv* env, jobject self, jlong bytes) {
  void* p = bytes > 0 ? calitas_alloc_host((uint64_t)bytes) : NULL;
  if (!p) { throw_state(env, "calitas_alloc_host failed"); return NULL; }
  return (*env)->NewDirectByteBuffer(env, p, bytes);
}

/* calitas_search_variants_into: searchVariants with the text delivered into `dst` (a direct ByteBuffer, ideally from allocHost); returns
 * the text's length in bytes.  vcfId may be null: the library then computes "name:md5" itself. */
JNIEXPORT jlong JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_searchVariantsInto(JNIEnv* env, jobject self, jlong h,
    jstring protospacer, jobjectArray pams, jboolean pam5, jint cliLength, jstring guideId, jintArray params, jstring vcfPath,
    jstring chrom, jstring vcfId, jstring version, jobject dst) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  calitas_guide_t g;
  calitas_params_t p;
  if (!read_params(env, params, &p)) return 0;
  char* out = dst ? (char*)(*env)->GetDirectBufferAddress(env, dst) : NULL;
  const jlong cap = dst ? (*env)->GetDirectBufferCapacity(env, dst) : 0;
  if (!out || cap < 2) { throw_state(env, "dst: a direct ByteBuffer is needed"); return 0; }
  const jsize np = (*env)->GetArrayLength(env, pams);
  const char* cp[CALITAS_MAX_PAMS];
  jstring jp[CALITAS_MAX_PAMS];
  if (np > CALITAS_MAX_PAMS) { throw_state(env, "too many PAMs"); return 0; }
  g.protospacer = (*env)->GetStringUTFChars(env, protospacer, NULL);
  for (jsize i = 0; i < np; i++) { jp[i] = (jstring)(*env)->GetObjectArrayElement(env, pams, i); cp[i] = (*env)->GetStringUTFChars(env, jp[i], NULL); }
  g.n_pams = (int32_t)np; g.pams = cp; g.pam_is_5prime = pam5 ? 1 : 0; g.cli_length = cliLength;
  const char* gid = (*env)->GetStringUTFChars(env, guideId, NULL);
  const char* vcf = (*env)->GetStringUTFChars(env, vcfPath, NULL);
  const char* vid = vcfId ? (*env)->GetStringUTFChars(env, vcfId, NULL) : NULL;
  const char* chr = chrom ? (*env)->GetStringUTFChars(env, chrom, NULL) : NULL;
  const char* ver = version ? (*env)->GetStringUTFChars(env, version, NULL) : NULL;
  uint64_t bytes = 0, rows = 0;
  const int rc = calitas_search_variants_into(ctx, &g, gid, &p, vcf, chr, vid, ver, NULL, out, (uint64_t)cap, &bytes, &rows, NULL);
  (*env)->ReleaseStringUTFChars(env, protospacer, g.protospacer);
  for (jsize i = 0; i < np; i++) (*env)->ReleaseStringUTFChars(env, jp[i], cp[i]);
  (*env)->ReleaseStringUTFChars(env, guideId, gid);
  (*env)->ReleaseStringUTFChars(env, vcfPath, vcf);
  if (vcfId) (*env)->ReleaseStringUTFChars(env, vcfId, vid);
  if (chrom) (*env)->ReleaseStringUTFChars(env, chrom, chr);
  if (version) (*env)->ReleaseStringUTFChars(env, version, ver);
  if (rc != CALITAS_OK) { throw_state(env, calitas_last_error(ctx)); return 0; }
  return (jlong)bytes;
}

/* calitas_vcf_identifier: ReferenceHit's "name:md5" of a VCF (ReferenceHit.scala:175-183), for callers that search many guides against
 * one VCF and pass it as vcfId. */
JNIEXPORT jstring JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_vcfIdentifier(JNIEnv* env, jobject self, jlong h, jstring vcfPath) {
  calitas_ctx* ctx = (calitas_ctx*)(intptr_t)h;
  const char* vcf = (*env)->GetStringUTFChars(env, vcfPath, NULL);
  char* id = NULL;
  const int rc = calitas_vcf_identifier(ctx, vcf, &id);
  (*env)->ReleaseStringUTFChars(env, vcfPath, vcf);
  if (rc != CALITAS_OK) { throw_state(env, calitas_last_error(ctx)); return NULL; }
  jstring out = (*env)->NewStringUTF(env, id);
  calitas_free(id);
  return out;
}

JNIEXPORT void JNICALL Java_com_editasmedicine_aligner_NativeAligner_00024_free(JNIEnv* env, jobject self, jobject buffer) {
  calitas_free((*env)->GetDirectBufferAddress(env, buffer));
}
