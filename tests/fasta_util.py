"""Writes small FASTA files with the .fai and .dict the reference requires (SR:479, RH:168), like fgbio's
ReferenceSetBuilder.toTempFile() does for the reference's own tests."""
import os


def expand(spec):
    """[[unit, repeat], ...] -> sequence string."""
    return "".join(u * n for u, n in spec)


def write_fasta(path, contigs, line_len=80, assembly="testassembly"):
    """contigs: list of (name, sequence). Writes path, path + '.fai' and <stem>.dict."""
    fai = []
    with open(path, "w") as f:
        for name, seq in contigs:
            f.write(">%s\n" % name)
            off = f.tell()
            for i in range(0, len(seq), line_len):
                f.write(seq[i:i + line_len] + "\n")
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, line_len, line_len + 1))
    with open(path + ".fai", "w") as f:
        f.writelines(fai)
    stem = os.path.splitext(path)[0]
    with open(stem + ".dict", "w") as f:
        f.write("@HD\tVN:1.5\tSO:unsorted\n")
        for name, seq in contigs:
            f.write("@SQ\tSN:%s\tLN:%d\tAS:%s\n" % (name, len(seq), assembly))
    return path
