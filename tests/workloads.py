"""Seeded scenarios shared by the CPU (oracle) and GPU (parity) tests."""
import numpy as np

from longreadmapper_amd import index, synth

_cache = {}


def repeat_reference(n_copies=250, elem_len=1000, spacer=600, seed=4):
    """An exact interspersed repeat: every 20-mer of the element occurs n_copies times."""
    rng = np.random.default_rng(seed)
    elem = synth.reference(elem_len, seed=seed + 100)
    parts = []
    for i in range(n_copies):
        parts.append(synth.reference(spacer + int(rng.integers(0, 40)), seed=seed * 1000 + i))
        parts.append(elem)
    parts.append(synth.reference(spacer, seed=seed * 1000 + n_copies))
    return np.concatenate(parts), elem


def ragged(reads, lens, new_lens):
    """Truncate reads in place to new lengths (NUL padded, like refactor_reads_seq)."""
    out = reads.copy()
    ln = np.array(new_lens, dtype=np.uint32)
    for i, l in enumerate(ln):
        out[i, l:] = 0
    return out, ln


def scenario(name):
    """-> dict(seqs, hi (HostIndex), reads (n, stride) uint8, lens, truth, seed_len, thres)"""
    if name in _cache:
        return _cache[name]
    if name == "clean-1k":            # BASELINE config 0 in small: clean-ish reads decide in phase 0
        seqs = [synth.reference(200_000, seed=1)]
        r = synth.reads(seqs, 96, 1000, dict(p_sub=0.01, p_ins=0.005, p_del=0.005), seed=11)
        sc = dict(seqs=seqs, seed_len=20, thres=300, hlen=8)
    elif name == "ont-2k":            # noisy reads: all phases run
        seqs = [synth.reference(150_000, seed=2), synth.reference(60_000, seed=3)]
        r = synth.reads(seqs, 64, 2000, synth.ONT, seed=13)
        sc = dict(seqs=seqs, seed_len=20, thres=300, hlen=8)
    elif name == "pacbio-3k-h12":     # hlen 12 like the reference's accidx
        seqs = [synth.reference(120_000, seed=5)]
        r = synth.reads(seqs, 32, 3000, synth.PACBIO_CLR, seed=17)
        sc = dict(seqs=seqs, seed_len=20, thres=300, hlen=12)
    elif name == "ragged":            # empty, shorter than a seed, exactly seed_len, seed_len+1, ...
        seqs = [synth.reference(80_000, seed=7)]
        r = synth.reads(seqs, 24, 700, synth.CLEAN, seed=19)
        newl = [0, 1, 19, 20, 21, 22, 40, 41, 42, 63, 64, 65, 100, 127, 128, 129, 250, 333, 500, 641, 699, 700, 700, 5]
        r["reads"], r["lens"] = ragged(r["reads"], r["lens"], newl)
        sc = dict(seqs=seqs, seed_len=20, thres=300, hlen=8)
    elif name == "repeats-ties":      # 40 exact copies, thres 50: every locus ties, first-seen order decides
        ref, elem = repeat_reference(40, 600, 300, seed=4)
        seqs = [ref]
        r = synth.reads(seqs, 48, 400, dict(p_sub=0.02, p_ins=0.0, p_del=0.0), seed=23)
        sc = dict(seqs=seqs, seed_len=20, thres=50, hlen=8)
    elif name == "repeats-overflow":  # 250 copies, thres 300: >192 distinct buckets per phase -> global vote table
        ref, elem = repeat_reference(250, 800, 500, seed=6)
        seqs = [ref]
        r = synth.reads(seqs, 40, 600, dict(p_sub=0.01, p_ins=0.005, p_del=0.005), seed=29)
        sc = dict(seqs=seqs, seed_len=20, thres=300, hlen=8)
    elif name == "seed12":            # seed_len == hlen: no backward steps at all
        seqs = [synth.reference(100_000, seed=8)]
        r = synth.reads(seqs, 32, 900, synth.ONT, seed=31)
        sc = dict(seqs=seqs, seed_len=12, thres=300, hlen=12)
    elif name == "seed16":            # shortest seed the seed table takes: cores of 26 / 30 bits, 6-byte slots with halfword tags
        seqs = [synth.reference(180_000, seed=14), synth.reference(30_000, seed=15)]
        r = synth.reads(seqs, 48, 1500, synth.ONT, seed=33)
        sc = dict(seqs=seqs, seed_len=16, thres=300, hlen=8)
    elif name == "seed24":            # longest seed the seed table takes: cores of 42 / 46 bits (the hash folds the bits above 32 in)
        seqs = [synth.reference(120_000, seed=16)]
        r = synth.reads(seqs, 40, 1200, dict(p_sub=0.02, p_ins=0.01, p_del=0.01), seed=35)
        sc = dict(seqs=seqs, seed_len=24, thres=300, hlen=8)
    elif name == "seed32":            # longest supported seed
        seqs = [synth.reference(100_000, seed=9)]
        r = synth.reads(seqs, 32, 1500, dict(p_sub=0.01, p_ins=0.01, p_del=0.01), seed=37)
        sc = dict(seqs=seqs, seed_len=32, thres=300, hlen=8)
    elif name == "seed-below-hlen":   # seed_len < hlen: lc_aln's else-branch (lchash.c:97-99), all seeds uninformative
        seqs = [synth.reference(30_000, seed=10)]
        r = synth.reads(seqs, 16, 300, synth.CLEAN, seed=41)
        sc = dict(seqs=seqs, seed_len=6, thres=300, hlen=8)
    elif name == "last-phase-break":  # read length tuned so that only a late phase can pass 0.6
        seqs = [synth.reference(50_000, seed=12)]
        r = synth.reads(seqs, 64, 62, synth.CLEAN, seed=43)       # num_seeds = 2: v >= 2 passes
        sc = dict(seqs=seqs, seed_len=20, thres=300, hlen=8)
    else:
        raise KeyError(name)
    sc["hi"] = index.HostIndex.build(sc["seqs"], o_ratio=32, hlen=sc["hlen"])
    sc.update(reads=r["reads"], lens=r["lens"], truth=r)
    _cache[name] = sc
    return sc


SEED_SCENARIOS = ["clean-1k", "ont-2k", "pacbio-3k-h12", "ragged", "repeats-ties", "repeats-overflow", "seed12",
                  "seed16", "seed24", "seed32", "seed-below-hlen", "last-phase-break"]
