"""Deterministic synthetic FASTA generator (SURVEY.md §8(d) "Synthetic inputs").

iid uniform A/C/G/T drawn from a splitmix64 stream (2 bits per base, low bits
first), upper case, records ``>p<i>`` / ``>n<i>`` written exactly like the
pipeline's own FASTA writer (reference scripts/seqs_nullgen.py:465 --
``">%s\\n%s\\n\\n"``: one sequence line followed by a blank line).

Pure numpy so that the development container and the GPU box generate the same
bytes without the reference being present.
"""
import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, count):
    """First `count` outputs of splitmix64(seed) as a uint64 array."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def random_bases(seed, nbases):
    """`nbases` base codes 0..3 (A,C,G,T) from splitmix64(seed)."""
    words = splitmix64(seed, (nbases + 31) // 32)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((words[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return codes[:nbases]


def random_lengths(seed, n, lo, hi):
    """n lengths iid uniform in [lo, hi] from splitmix64(seed)."""
    r = splitmix64(seed, n)
    return (lo + (r % np.uint64(hi - lo + 1))).astype(np.int64)


def make_sequences(seed, n, length=300, length_range=None, length_seed=3):
    """List of n byte strings.  Fixed `length`, or lengths uniform in length_range."""
    if length_range is None:
        lens = np.full(n, length, dtype=np.int64)
    else:
        lens = random_lengths(length_seed + 1000 * seed, n, length_range[0], length_range[1])
    codes = random_bases(seed, int(lens.sum()))
    letters = _BASES[codes].tobytes()
    out, off = [], 0
    for ln in lens:
        out.append(letters[off:off + int(ln)])
        off += int(ln)
    return out


def write_fasta(path, seqs, prefix):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">" + prefix.encode() + str(i).encode() + b"\n" + s + b"\n\n")


def write_problem(pos_path, neg_path, n_pos, n_neg, length=300, length_range=None,
                  seed_pos=1, seed_neg=2):
    """Write the positive and negative FASTA files of a synthetic problem."""
    pos = make_sequences(seed_pos, n_pos, length, length_range)
    neg = make_sequences(seed_neg, n_neg, length, length_range)
    write_fasta(pos_path, pos, "p")
    write_fasta(neg_path, neg, "n")
    return pos, neg


# ---------------------------------------------------------------------- peak-like sequences
# A stand-in for what `bin/gkmqc.py evaluate` really feeds the kernel (reference bin/gkmqc.py:150-154,
# 181-185, 338-343: subsets of 5 000 peak windows of 600 bp around the summit + 5 000 GC-/repeat-
# matched null windows, L=10 k=6 d=3): the genome is not available here, so the generator plants
# the features that make real peaks differ from iid ACGT for THIS kernel -- shared l-mers:
#   * per-sequence GC content (regulatory DNA is GC-skewed, peak to peak)
#   * motif families concentrated near the summit (positives only)
#   * poly-A / poly-T tracts and (AC)n-style dinucleotide repeats
#   * fragments of one shared Alu-like element, either strand, 10 % diverged
#   * a few N (preprocessing lets peaks with <= 1 % N through, scripts/preprocess.py:119)
# Everything is drawn from splitmix64 streams, so both boxes produce the same bytes.

_PEAK_MOTIFS = (b"TGACTCAGCA", b"GGGCGGGGCC", b"CCACGTGGTC", b"GATAAGATCT", b"TTGCGCAATA", b"CAGCTGTTCC",
                b"GGAAGTGACG", b"TGTTTACTTA")


class _Stream:
    """Uniform variates from one splitmix64 stream, served in blocks."""

    def __init__(self, seed, block=1 << 16):
        self.seed, self.block, self.served = int(seed), block, 0
        self.buf = np.zeros(0, dtype=np.float64)

    def uniform(self, count):
        while len(self.buf) < count:
            lo = self.served
            idx = np.arange(lo + 1, lo + self.block + 1, dtype=np.uint64)
            with np.errstate(over="ignore"):
                z = (np.uint64(self.seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & _M64
                z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
                z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
                z = z ^ (z >> np.uint64(31))
            self.buf = np.concatenate([self.buf, (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)])
            self.served += self.block
        out, self.buf = self.buf[:count], self.buf[count:]
        return out

    def one(self):
        return float(self.uniform(1)[0])

    def integer(self, lo, hi):
        """uniform in [lo, hi]"""
        return lo + int(self.one() * (hi - lo + 1))


def _bases_with_gc(u, gc):
    edges = np.array([(1 - gc) / 2, (1 - gc) / 2 + gc / 2, (1 - gc) / 2 + gc])
    return np.searchsorted(edges, u, side="right").astype(np.uint8)       # 0..3 = A, C, G, T


def _revcomp_codes(codes):
    return (3 - codes)[::-1]


def make_peak_sequences(seed, n, length=600, positives=True):
    """n byte strings of peak-like DNA (see above).  positives=False: the matched null set (same
    composition and repeat content, no motifs)."""
    rs = _Stream(0x5EED0000 + int(seed))
    alu = _bases_with_gc(_Stream(0xA1A1).uniform(280), 0.56)              # ONE shared element for all sets
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for _ in range(n):
        z = float(rs.uniform(4).sum() - 2.0) * 1.7                         # ~N(0,1)
        gc = min(0.70, max(0.30, 0.46 + 0.08 * z))
        codes = _bases_with_gc(rs.uniform(length), gc)
        if rs.one() < 0.10:                                                # Alu-like fragment
            flen = rs.integer(100, 280)
            start = rs.integer(0, 280 - flen)
            frag = alu[start:start + flen].copy()
            mut = rs.uniform(flen) < 0.10
            frag[mut] = (frag[mut] + 1 + (rs.uniform(int(mut.sum())) * 3).astype(np.uint8)) & 3
            if rs.one() < 0.5:
                frag = _revcomp_codes(frag)
            at = rs.integer(0, length - flen)
            codes[at:at + flen] = frag
        if rs.one() < 0.15:                                                # poly-A / poly-T tract
            run = rs.integer(10, 30)
            at = rs.integer(0, length - run)
            codes[at:at + run] = 0 if rs.one() < 0.5 else 3
        if rs.one() < 0.08:                                                # dinucleotide repeat
            unit = [(0, 1), (3, 2), (1, 0), (2, 3), (0, 2), (3, 1)][rs.integer(0, 5)]
            run = rs.integer(20, 50)
            at = rs.integer(0, length - run)
            codes[at:at + run] = np.resize(np.array(unit, dtype=np.uint8), run)
        if positives:
            for _m in range(rs.integer(1, 3)):
                motif = np.frombuffer(_PEAK_MOTIFS[rs.integer(0, len(_PEAK_MOTIFS) - 1)], dtype=np.uint8)
                m = np.searchsorted(letters, motif).astype(np.uint8)
                mut = rs.uniform(len(m)) < 0.12
                m = m.copy()
                m[mut] = (m[mut] + 1 + (rs.uniform(int(mut.sum())) * 3).astype(np.uint8)) & 3
                if rs.one() < 0.5:
                    m = _revcomp_codes(m)
                zc = float(rs.uniform(4).sum() - 2.0) * 1.7
                at = int(min(length - len(m), max(0, length // 2 + 40 * zc - len(m) // 2)))
                codes[at:at + len(m)] = m
        s = bytearray(letters[codes].tobytes())
        if rs.one() < 0.02:                                                # a few N (read as A by the kernel)
            for _k in range(rs.integer(1, 6)):
                s[rs.integer(0, length - 1)] = ord("N")
        out.append(bytes(s))
    return out


def write_peak_problem(pos_path, neg_path, n_pos, n_neg, length=600, seed_pos=11, seed_neg=12):
    """The peak-like stand-in for one `gkmqc.py evaluate` subset: positives + matched nulls."""
    pos = make_peak_sequences(seed_pos, n_pos, length, True)
    neg = make_peak_sequences(seed_neg, n_neg, length, False)
    write_fasta(pos_path, pos, "p")
    write_fasta(neg_path, neg, "n")
    return pos, neg
