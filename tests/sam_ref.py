"""Python restatement of the reference's SAM text (alnmain.c:62-75 header, :485-527 records), used by
the tests to check lrm_sam_header / lrm_sam_format / lrm_accaln.  Unmapped reads are fenced (the
reference prints an uninitialised struct): RNAME "*", POS 0, CIGAR "*"."""


def rle(ops: bytes) -> str:
    if not ops:
        return "*"
    out, i = [], 0
    while i < len(ops):
        o = "M" if ops[i:i + 1] in (b"=", b"X") else chr(ops[i])
        j = i
        while j < len(ops) and ("M" if ops[j:j + 1] in (b"=", b"X") else chr(ops[j])) == o:
            j += 1
        out.append("%d%s" % (j - i, o))
        i = j
    return "".join(out)


def header(mta, rg_id):
    s = "".join("@SQ\tSN:%s\tLN:%d\n" % (name, ln) for name, _, ln in mta)
    return s + "@RG\tID:accaln%d\tSM:SM_data\n" % rg_id + "@PG\tID:accaln\tPN:accaln\n"


def record(name, seq, qual, mta, ops, score, meta_r, seq_id, off, strand):
    unmapped = meta_r == 0 or score == -1
    flag = 4 if unmapped else (16 if strand == 1 else 0)
    mapq = 0 if unmapped else 255
    return "%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tED:I:%d\n" % (
        name, flag, "*" if unmapped else mta[seq_id][0], 0 if unmapped else off + 1, mapq,
        "*" if unmapped else rle(ops), seq, qual if qual is not None else "*", score)
