"""NumPy restatement of the device noise generator -- TEST INFRASTRUCTURE ONLY.

The reference draws eps with TensorFlow's stateful Philox stream through TFP's seed splitting
(src/iwae1.py:59); that stream cannot be reproduced without TensorFlow, so bit-parity of the noise
is out of reach by construction ("RNG parity").  The device generator is the published
Philox4x32-10 counter RNG (Salmon et al., SC'11; Random123 known-answer vectors are checked in
tests/test_oracle.py) + Box-Muller, keyed so that a data row's noise depends only on
(seed, step, global row index, feature) -- never on the batch split or the GPU count:

    counter = (row_lo, row_hi, (stream << 24) | d4, step), key = (seed_lo, seed_hi)
    u_i = ((r_i >> 8) + 0.5) * 2^-24 ;  n0,n1 = sqrt(-2 ln u0) * (cos, sin)(2 pi u1) ; n2,n3 likewise from u2,u3
    eps[row, 4*d4 + j] = n_j
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over arrays of uint32 counters; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def device_eps(seed, step, B, k, D, stream=0, batch_offset=0):
    """The N(0,1) draws the device produces for a [B images, k samples, D features] call, in the
    reference's [k, B, D] order (float64; the device evaluates the same formula in float32)."""
    nd4 = (D + 3) // 4
    b = np.arange(B)[:, None, None]
    s = np.arange(k)[None, :, None]
    d4 = np.arange(nd4)[None, None, :]
    grow = (np.uint64(batch_offset) * np.uint64(k) + (b * k + s).astype(np.uint64)) + np.zeros((B, k, nd4), dtype=np.uint64)
    c0 = (grow & MASK).astype(np.uint32)
    c1 = (grow >> np.uint64(32)).astype(np.uint32)
    c2 = ((np.uint32(stream) << np.uint32(24)) | d4.astype(np.uint32)) + np.zeros((B, k, nd4), dtype=np.uint32)
    c3 = np.full((B, k, nd4), step, dtype=np.uint32)
    r = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = [((x >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24 for x in r]
    ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    n = np.stack([ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]),
                  rb * np.cos(2 * np.pi * u[3]), rb * np.sin(2 * np.pi * u[3])], axis=-1)   # [B,k,nd4,4]
    n = n.reshape(B, k, nd4 * 4)[:, :, :D]
    return n.transpose(1, 0, 2)


def device_binarize(seed, epoch, gray_u8, image_ids):
    """Bit-exact restatement of gather_binarize_kernel: rows = images `image_ids` of the uint8 dataset,
    x = 1 iff (philox(counter = (image, 'BINA', pixel/4, epoch), key = seed)[pixel%4] >> 8) < floor(g*2^24/255 + 0.5)
    (the reference's per-epoch np.random.binomial(1, g/255), src/utils.py:26-27, with an integer threshold)."""
    gray_u8 = np.asarray(gray_u8, dtype=np.uint8)
    ids = np.asarray(image_ids, dtype=np.int64)
    X = gray_u8.shape[1]
    nd4 = (X + 3) // 4
    c0 = np.broadcast_to(ids[:, None].astype(np.uint32), (ids.size, nd4))
    c1 = np.full((ids.size, nd4), 0x42494E41, dtype=np.uint32)
    c2 = np.broadcast_to(np.arange(nd4, dtype=np.uint32)[None, :], (ids.size, nd4))
    c3 = np.full((ids.size, nd4), epoch, dtype=np.uint32)
    r = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack([x >> np.uint32(8) for x in r], axis=-1).reshape(ids.size, nd4 * 4)[:, :X].astype(np.uint64)
    g = gray_u8[ids].astype(np.uint64)
    thr = (g * np.uint64(16777216 * 2) + np.uint64(255)) // np.uint64(510)
    return (u < thr).astype(np.float32)
