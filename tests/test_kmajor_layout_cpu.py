"""CPU model of the K-major operand path of csrc/gemm_kmajor.hip (no GPU): the LDS image the LDS-DMA lane assignment
produces, the address every lane hands to ds_read_b64_tr_b16 (semantics as probed on the chip by tools/tr_probe.hip:
lane i of a 16-lane group receives element i % 4 of the 8 bytes addressed by lanes i / 4 + {0, 4, 8, 12}), and the
32x32x16 MFMA operand layout (lane l: row l & 31, k = 8 (l >> 5) .. + 7).  Guards the index arithmetic of the kernel:
every lane must end up with A[k][m] for ITS row and k range, and a 32-lane pass must touch 64 distinct LDS banks."""
import numpy as np

PIECE, NPL = 1024, 3
GROUP = NPL * PIECE + 128


def lds_image(tile, plane_values):
    """tile: [32 k][128 m] array of 16-bit ids for one plane -> byte-addressed LDS region (dict addr -> id per 2 B)"""
    lds = {}
    for j in range(8):                                   # 16-column group
        for lane in range(64):                           # one LDS-DMA instruction: lane -> (k-row lane >> 1, piece lane & 1)
            k, piece = lane >> 1, lane & 1
            base = j * GROUP + plane_values * PIECE + lane * 16
            for e in range(8):                           # 8 columns of the piece, 2 bytes each
                lds[base + 2 * e] = tile[k, j * 16 + piece * 8 + e]
    return lds


def tr_read(lds, addr_of_lane):
    """the probed semantics: out[i][j] = element (i % 4) of the 8 bytes addressed by lane 4 j + i / 4 of i's group"""
    out = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for j in range(4):
            src = g * 16 + 4 * j + i // 4
            out[lane, j] = lds[addr_of_lane[src] + 2 * (i % 4)]
    return out


def test_kmajor_fragments_are_the_mfma_operand_and_conflict_free():
    rng = np.random.default_rng(0)
    tile = rng.permutation(32 * 128).reshape(32, 128)    # unique id per (k, m)
    for plane in range(NPL):
        lds = lds_image(tile, plane)
        for w in range(2):                               # wave row (64 output rows each)
            for i in range(2):                           # 32-row tile of the wave
                for ks in range(2):                      # 16-k step
                    frag = np.zeros((64, 8), dtype=np.int64)
                    for rd in range(2):                  # two reads: k .. k+3, k+4 .. k+7
                        addr = []
                        for lane in range(64):
                            g16, h, s = (lane >> 4) & 1, lane >> 5, lane & 15
                            a = (w * 4 + 2 * i + g16) * GROUP + plane * PIECE + (16 * ks + 8 * h + (s >> 2)) * 32 \
                                + (s & 3) * 8 + rd * 128
                            assert a % 8 == 0            # a misaligned tr read returns the aligned address's data
                            addr.append(a)
                        # banks of a 32-lane pass: 8 bytes = 2 dwords per lane, 64 banks of 4 B
                        for half in range(2):
                            banks = set()
                            for lane in range(32 * half, 32 * half + 32):
                                banks.update({(addr[lane] // 4) % 64, (addr[lane] // 4 + 1) % 64})
                            assert len(banks) == 64
                        frag[:, 4 * rd:4 * rd + 4] = tr_read(lds, addr)
                    for lane in range(64):
                        m = w * 64 + i * 32 + (lane & 31)
                        k0 = 16 * ks + 8 * (lane >> 5)
                        assert list(frag[lane]) == [tile[k0 + e, m] for e in range(8)], (plane, w, i, ks, lane)
