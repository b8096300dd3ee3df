"""CPU restatement of the engine's Gaussian stream -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench's cpu leg).

The reference draws Omega with rand 0.8 + rand_distr 0.4 (ziggurat) from a caller-owned RNG
(/root/reference/src/random_matrix.rs:120-125: `Normal::new(0.0, 1.0)`, sampled in f64, cast to T,
filled in row-major order).  Those crates are not vendored and carry no pinned version (no Cargo.lock), and a
GPU cannot consume a sequential host RNG, so the engine defines its own stream in its ABI
(include/rusty_compression_amd.h, "random_matrix.rs" section): Philox4x32-10 (Salmon, Moraes, Dror, Shaw:
"Parallel random numbers: as easy as 1, 2, 3", SC'11) + Box-Muller.  The sketch stage is therefore
"parity unpinned by the reference" (it has no tests there either); what IS pinned here:

  * the integer generator against Random123's published known-answer vectors (kat_vectors, philox4x32 10 rounds),
  * the GPU kernel `k_fill_gaussian` against this file: uint32 words bit-exact, normals to a few ulp.

Stream contract restated from the header:
  block b (a 64-bit counter, little end first, upper counter words 0) with key (seed_lo, seed_hi) gives 4 words w0..w3;
  a = w0 << 32 | w1, b = w2 << 32 | w3;  u1 = ((a >> 11) + 1) * 2^-53 in (0, 1],  u2 = (b >> 11) * 2^-53 in [0, 1);
  z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = sqrt(-2 ln u1) sin(2 pi u2);
  normal number e of the stream (seed, offset) is z_{(offset+e) & 1} of block (offset + e) >> 1;
  element (i, j) of an r x c matrix is number i * c + j (row-major draw order, as the reference fills); f32 = cast of f64.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """ctr: (n, 4) uint32 counters, key: (k0, k1).  Returns (n, 4) uint32.  Pure integer arithmetic."""
    c = np.asarray(ctr, dtype=np.uint64).copy()
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[:, 0]            # 32 x 32 -> 64 bit products (operands < 2^32: no overflow in uint64)
        p1 = M1 * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c[:, 1] ^ np.uint64(k0)
        n2 = hi0 ^ c[:, 3] ^ np.uint64(k1)
        c = np.stack([n0, lo1, n2, lo0], axis=1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def blocks(seed: int, first_block: int, nblocks: int) -> np.ndarray:
    """Words of blocks first_block .. first_block + nblocks - 1 of the stream keyed by `seed`: (nblocks, 4) uint32."""
    b = np.uint64(first_block) + np.arange(nblocks, dtype=np.uint64)
    ctr = np.zeros((nblocks, 4), dtype=np.uint64)
    ctr[:, 0] = b & MASK
    ctr[:, 1] = b >> np.uint64(32)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))


def words(seed: int, word_offset: int, n: int) -> np.ndarray:
    """n consecutive uint32 words of the stream starting at word number `word_offset` (word w = w_{w & 3} of block w >> 2)."""
    first = word_offset >> 2
    nb = ((word_offset + n + 3) >> 2) - first
    flat = blocks(seed, first, nb).reshape(-1)
    s = word_offset - 4 * first
    return flat[s:s + n].copy()


def normals(seed: int, offset: int, n: int) -> np.ndarray:
    """n consecutive N(0,1) numbers (f64) of the stream (seed, offset); evaluated in extended precision and rounded
    once, so each value is the correctly rounded Box-Muller image of its two uniforms to ~0.5 ulp."""
    first = offset >> 1
    nb = ((offset + n + 1) >> 1) - first
    w = blocks(seed, first, nb).astype(np.uint64)
    a = (w[:, 0] << np.uint64(32)) | w[:, 1]
    b = (w[:, 2] << np.uint64(32)) | w[:, 3]
    ld = np.longdouble
    u1 = ((a >> np.uint64(11)).astype(ld) + ld(1)) * ld(2.0) ** -53
    u2 = (b >> np.uint64(11)).astype(ld) * ld(2.0) ** -53
    rad = np.sqrt(ld(-2) * np.log(u1))
    # cos / sin of 2 pi u2 through the quadrant-reduced argument (exact reduction: u2 is a dyadic rational)
    ang = ld(2) * u2                     # in [0, 2): multiples of pi
    two_pi = ld(2) * np.arccos(ld(-1))
    z = np.empty((nb, 2), dtype=ld)
    z[:, 0] = rad * np.cos(two_pi * u2)
    z[:, 1] = rad * np.sin(two_pi * u2)
    # exact zeros / units where the angle is a multiple of pi / 2 (sincospi semantics)
    for q, (c, s) in {0.0: (1, 0), 0.5: (0, 1), 1.0: (-1, 0), 1.5: (0, -1)}.items():
        hit = ang == ld(q)
        z[hit, 0] = rad[hit] * c
        z[hit, 1] = rad[hit] * s
    flat = z.reshape(-1)
    s0 = offset - 2 * first
    return flat[s0:s0 + n].astype(np.float64)


def random_gaussian(shape, seed: int, offset: int = 0, dtype=np.float64) -> np.ndarray:
    """The matrix rc_random_gaussian_{f64,f32}(out[rows x cols], seed, offset) fills: row-major draw order, drawn in f64, cast."""
    r, c = shape
    return normals(seed, offset, r * c).reshape(r, c).astype(dtype)


# Random123 known-answer vectors for philox4x32, 10 rounds (Random123 distribution, examples/kat_vectors):
#   counter words, key words -> output words
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
