"""TEST INFRASTRUCTURE ONLY — CPU restatement of the dropout keep-mask of csrc/norm.hip.

The reference's nn.Dropout (src/module.py:118-119,137-138; src/asr.py:36,162) draws from torch's
CUDA Philox stream, which no other implementation can reproduce bit-for-bit; what CAN be pinned is
that the kernel's mask is exactly the documented pure function of (seed, offset, index):
word (i % 4) of Philox4x32-10(counter = (i // 4, offset), key = seed), keep <=> (word >> 8) >=
round(p * 2^24)  (include/asrk.h).  Philox4x32-10 follows Salmon et al., "Parallel Random Numbers:
As Easy as 1, 2, 3" (SC'11) and is checked below against the paper's known-answer vectors.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [n,4], key: (k0, k1) python ints -> uint32 [n,4]"""
    c = [ctr[:, j].astype(np.uint64) for j in range(4)]
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & MASK32
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & MASK32
        c = [n0, p1 & MASK32, n2, p0 & MASK32]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def keep_mask(n, p, seed, offset=0):
    """bool [n]: which elements inverted dropout keeps"""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    ctr = np.stack([q & MASK32, q >> np.uint64(32),
                    np.full(nq, offset & 0xFFFFFFFF, np.uint64), np.full(nq, offset >> 32, np.uint64)],
                   axis=1).astype(np.uint32)
    words = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).reshape(-1)[:n]
    thresh = np.uint32(int(np.rint(np.float32(p) * np.float32(16777216.0))))
    return (words >> np.uint32(8)) >= thresh


def dropout(x, p, seed, offset=0):
    """float32 numpy inverted dropout with the kernel's mask and scale"""
    x = np.asarray(x, np.float32)
    keep = keep_mask(x.size, p, seed, offset).reshape(x.shape)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(keep, x * scale, np.float32(0.0)).astype(np.float32)


# Known-answer vectors of Random123's kat_vectors for philox4x32-10
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
