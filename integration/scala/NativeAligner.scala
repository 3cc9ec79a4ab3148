package com.editasmedicine.aligner

import java.nio.{ByteBuffer, ByteOrder}

import com.editasmedicine.aligner.SequentialGuideAligner.Guide
import com.fulcrumgenomics.alignment.Cigar

/** JNI wrapper over libcalitas_hip.so (include/calitas_hip.h): one instance = one MI355X = one calitas_ctx.
  * NOT compiled in the calitas-mi355x repository (no JDK there); drop into calitas/src/main/scala/com/editasmedicine/aligner/
  * next to SearchReference.scala and build integration/jni/calitas_jni.c into libcalitas_jni.so. */
final class NativeAligner(device: Int = 0) extends AutoCloseable {
  System.loadLibrary("calitas_jni")
  private val handle: Long = NativeAligner.create(device)

  /** calitas_set_reference: contigs in sequence-dictionary order with the bytes windowIterator sees (SearchReference.scala:41-49). */
  def setReference(names: Array[String], bases: Array[Array[Byte]], genomeBuild: String): Unit =
    NativeAligner.setReference(handle, names, bases, genomeBuild)

  /** The same for one process of a multi-GPU job (round 5): every contig of the sequence dictionary with its length, `null` bases for
    * the contigs this process's window range does not touch -- names, lengths, windowIterator's sequence and every coordinate stay
    * those of the whole dictionary. */
  def setReference(names: Array[String], lengths: Array[Long], bases: Array[Array[Byte]], genomeBuild: String): Unit =
    NativeAligner.setReferenceWithLengths(handle, names, lengths, bases, genomeBuild)

  /** calitas_search for one guide; returns the GuideAlignments of every window in windowIterator order. */
  def search(guide: Guide, cliLength: Int, params: Array[Int], contigNames: IndexedSeq[String],
             fetch: (String, Int, Int) => Array[Byte]): IndexedSeq[GuideAlignment] = {
    val pams = (guide.pams5Prime ++ guide.pams3Prime).toArray
    val buf  = NativeAligner.search(handle, guide.guide, pams, guide.pamIsFivePrime, cliLength, params).order(ByteOrder.LITTLE_ENDIAN)
    try NativeAligner.decode(buf, guide, pams, contigNames, fetch) finally NativeAligner.free(buf)
  }

  /** calitas_search_hits: the whole reference-genome branch of SearchReference.execute (SearchReference.scala:527-564, 641-648)
    * for one guide -- per-window filter, removeOverlaps, ReferenceHit.sort and the rows are done on the GPU; the bytes returned
    * are the hits.txt Metric.writer would have written (header + rows). */
  def searchHits(guide: Guide, cliLength: Int, guideId: String, params: Array[Int], version: String): Array[Byte] = {
    val pams = (guide.pams5Prime ++ guide.pams3Prime).toArray
    val buf  = NativeAligner.searchHits(handle, guide.guide, pams, guide.pamIsFivePrime, cliLength, guideId, params, version)
    try { val out = new Array[Byte](buf.capacity()); buf.get(out); out } finally NativeAligner.free(buf)
  }

  /** calitas_search_variants: SearchReference.execute with --variants (SearchReference.scala:570-648) -- variant windows, their
    * alignment, the lift-back to reference coordinates and the merge with the reference hits.  vcfId = ReferenceHit's
    * "name:md5" identifier (ReferenceHit.scala:175-183); params carries --max-variants. */
  def searchVariants(guide: Guide, cliLength: Int, guideId: String, params: Array[Int], vcf: java.nio.file.Path, chrom: Option[String],
                     vcfId: String, version: String): Array[Byte] = {
    val pams = (guide.pams5Prime ++ guide.pams3Prime).toArray
    val buf  = NativeAligner.searchVariants(handle, guide.guide, pams, guide.pamIsFivePrime, cliLength, guideId, params, vcf.toString,
                                            chrom.orNull, vcfId, version)
    try { val out = new Array[Byte](buf.capacity()); buf.get(out); out } finally NativeAligner.free(buf)
  }

  /** calitas_search_variants_into (round 5): the same search with hits.txt delivered into `dst` -- a direct buffer, ideally one from
    * NativeAligner.allocHost (page-locked memory of the runtime's own, reused from call to call): every contig's rows then cross the
    * bus once, straight to their place (0.55 s per whole-genome call against 0.9 s into a block of the library's).  Returns the
    * text's length; SearchReference.scala:641-648 writes dst.limit(n) to --output instead of going through Metric.write.
    * vcfId = None: the library computes "name:md5" itself (0.14 s per 127 MB, per call); vcfIdentifier below computes it once. */
  def searchVariantsInto(guide: Guide, cliLength: Int, guideId: String, params: Array[Int], vcf: java.nio.file.Path, chrom: Option[String],
                         vcfId: Option[String], version: String, dst: ByteBuffer): Long = {
    val pams = (guide.pams5Prime ++ guide.pams3Prime).toArray
    NativeAligner.searchVariantsInto(handle, guide.guide, pams, guide.pamIsFivePrime, cliLength, guideId, params, vcf.toString, chrom.orNull,
                                     vcfId.orNull, version, dst)
  }

  /** calitas_vcf_identifier: ReferenceHit's "name:md5" (ReferenceHit.scala:175-183) as the library computes it. */
  def vcfIdentifier(vcf: java.nio.file.Path): String = NativeAligner.vcfIdentifier(handle, vcf.toString)

  override def close(): Unit = NativeAligner.destroy(handle)
}

object NativeAligner {
  private val RecordBytes = 168 // sizeof(calitas_aln_t): 8 x int32, 2 x int8, int16, 128 op bytes
  private val MaxOps      = 128

  @native private def create(device: Int): Long
  @native private def destroy(handle: Long): Unit
  @native private def setReference(handle: Long, names: Array[String], bases: Array[Array[Byte]], genomeBuild: String): Unit
  @native private def setReferenceWithLengths(handle: Long, names: Array[String], lengths: Array[Long], bases: Array[Array[Byte]],
                                              genomeBuild: String): Unit
  @native private def search(handle: Long, protospacer: String, pams: Array[String], pamIsFivePrime: Boolean, cliLength: Int,
                             params: Array[Int]): ByteBuffer
  @native private def searchHits(handle: Long, protospacer: String, pams: Array[String], pamIsFivePrime: Boolean, cliLength: Int,
                                 guideId: String, params: Array[Int], version: String): ByteBuffer
  @native private def searchVariants(handle: Long, protospacer: String, pams: Array[String], pamIsFivePrime: Boolean, cliLength: Int,
                                     guideId: String, params: Array[Int], vcfPath: String, chrom: String, vcfId: String,
                                     version: String): ByteBuffer
  @native private def searchVariantsInto(handle: Long, protospacer: String, pams: Array[String], pamIsFivePrime: Boolean, cliLength: Int,
                                         guideId: String, params: Array[Int], vcfPath: String, chrom: String, vcfId: String,
                                         version: String, dst: ByteBuffer): Long
  @native private def vcfIdentifier(handle: Long, vcfPath: String): String
  /** calitas_alloc_host: a page-locked block as a direct buffer; release it with NativeAligner.release. */
  @native def allocHost(bytes: Long): ByteBuffer
  def release(buffer: ByteBuffer): Unit = free(buffer)
  @native private def free(buffer: ByteBuffer): Unit

  /** calitas_aln_t -> GuideAlignment (GuideAlignment.scala:72-88).  The padded strings follow Alignment.paddedString as used at
    * SequentialGuideAligner.scala:511: one op byte per padded column, already in guide orientation. */
  private def decode(buf: ByteBuffer, guide: Guide, pams: Array[String], contigs: IndexedSeq[String],
                     fetch: (String, Int, Int) => Array[Byte]): IndexedSeq[GuideAlignment] = {
    val n = buf.capacity() / RecordBytes
    Range(0, n).map { i =>
      val o        = i * RecordBytes
      val chrom    = contigs(buf.getInt(o + 4))
      val start    = buf.getInt(o + 12); val end    = buf.getInt(o + 16)
      val gStart   = buf.getInt(o + 20); val gEnd   = buf.getInt(o + 24)
      val score    = buf.getInt(o + 28)
      val strand   = buf.get(o + 32).toChar
      val pamIndex = buf.get(o + 33).toInt
      val nOps     = buf.getShort(o + 34).toInt
      val ops      = Array.tabulate(nOps)(k => buf.get(o + 36 + k).toChar)
      val pam      = if (pamIndex >= 0) pams(pamIndex) else ""
      val query    = if (guide.pamIsFivePrime) pam + guide.guide else guide.guide + pam
      val fwd      = new String(fetch(chrom, start, end)).toUpperCase
      val target   = if (strand == '-') com.fulcrumgenomics.util.Sequences.revcomp(fwd) else fwd
      val (pg, pa, pt) = (new StringBuilder, new StringBuilder, new StringBuilder)
      var (qi, ti) = (0, 0)
      ops.foreach {
        case 'I' => pg += query(qi); pa += '~'; pt += '-'; qi += 1
        case 'D' => pg += '-'; pa += '~'; pt += target(ti); ti += 1
        case '=' => pg += query(qi); pa += '|'; pt += target(ti); qi += 1; ti += 1
        case _   => pg += query(qi); pa += '.'; pt += target(ti); qi += 1; ti += 1
      }
      val cigar = Cigar(ops.map(_.toString).mkString.replaceAll("(.)", "1$1")).coalesce
      GuideAlignment(guide = query, chrom = chrom, startOffset = start, endOffset = end, guideStartOffset = gStart, guideEndOffset = gEnd,
        strand = strand, score = score, cigar = cigar, paddedGuide = pg.result(), paddedAlignment = pa.result(), paddedTarget = pt.result())
    }
  }
}
