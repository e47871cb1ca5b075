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
