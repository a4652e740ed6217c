"""Jump-ahead for numpy's MT19937 (`np.random.RandomState`), so that the reference's noise stream (augment/unet3d_augment/transforms.py:608-619:
`random_state.normal(0, std, size=m.shape)`) can be produced by MANY workgroups at once instead of one (csrc/mt19937.hip).

The generator is linear over GF(2): with F = "advance the 19937-bit state by one word", the word sequence x_k satisfies  sum_i phi_i x_{k+i} = 0  for the
characteristic polynomial phi (degree 19937, primitive) of F, for every k >= 1 (x_0's low 31 bits are not part of the state).  So if g(t) = t^(J-1) mod phi(t), then
x_{J-1+j} = XOR over {i : g_i = 1} of x_{i+j} for j >= 1: the 624 words of the key J steps ahead are a GF(2) "convolution" of the next ~20.6 k words of the
stream with the bit mask g - embarrassingly parallel over the 624 outputs (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer, "Efficient jump ahead for
F2-linear random number generators", 2008; evaluated here in its plain polynomial form, which is what fits one workgroup with the stream in LDS).

This module is host-side integer arithmetic only (Python ints as GF(2)[t] polynomials, bit i = coefficient of t^i):
  phi()                 the characteristic polynomial, from Berlekamp-Massey on 2 x 19937 output bits of any numpy stream (once per process, ~0.3 s)
  jump_polys(J, levels) [t^(J * 2^k - 1) mod phi for k < levels] as uint32 word arrays for the device (cached per J)
  jump_polys_from_start(J, n)  [t^(c J - 1) mod phi for c = 1 .. n-1]: every chunk key straight from the first one, one level of independent jumps
  jump_key(key, g)      reference implementation of the jump in numpy (tests; the device kernel mt_jump_kernel does the same sums)
"""
import numpy as np

DEG = 19937
N = 624
_PHI = None
_CACHE = {}


def _berlekamp_massey(bits):
    """connection polynomial C (int, bit i = c_i, c_0 = 1) of the shortest LFSR generating `bits`: s_n = XOR_{i=1..L} c_i s_{n-i}; returns (C, L)"""
    C, B, L, m = 1, 1, 0, 1
    R = 0                                   # reversed prefix: bit i = s_{n-i}
    for n, s in enumerate(bits):
        R = (R << 1) | s
        if (C & R).bit_count() & 1:         # discrepancy
            T = C
            C ^= B << m
            if 2 * L <= n:
                L, B, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
    return C, L


def phi():
    """characteristic polynomial of MT19937's one-word transition, as an int (bit i = coefficient of t^i, degree 19937)"""
    global _PHI
    if _PHI is None:
        rs = np.random.RandomState(12345)
        words = rs.randint(0, 2 ** 32, size=2 * DEG + 64, dtype=np.uint64)      # tempering is linear: bit 0 of the OUTPUT words satisfies the same recurrence
        C, L = _berlekamp_massey([int(w) & 1 for w in words])
        if L != DEG:
            raise RuntimeError(f"Berlekamp-Massey found a recurrence of length {L}, expected {DEG}")
        p = 0
        for i in range(L + 1):              # phi(t) = t^L C(1/t)
            if (C >> i) & 1:
                p |= 1 << (L - i)
        _PHI = p
    return _PHI


_SPREAD = None


def _square(a):
    """a(t)^2 over GF(2): spread the bits (bit i -> bit 2i)"""
    global _SPREAD
    if _SPREAD is None:
        t = np.zeros(256, dtype=np.uint16)
        for b in range(256):
            v = 0
            for i in range(8):
                if (b >> i) & 1:
                    v |= 1 << (2 * i)
            t[b] = v
        _SPREAD = t
    nbytes = (a.bit_length() + 7) // 8
    by = np.frombuffer(a.to_bytes(max(nbytes, 1), "little"), dtype=np.uint8)
    return int.from_bytes(_SPREAD[by].astype("<u2").tobytes(), "little")


def _mod(a, p):
    """a mod p over GF(2)"""
    dp = p.bit_length() - 1
    while True:
        d = a.bit_length() - 1
        if d < dp:
            return a
        a ^= p << (d - dp)


def _xpow(e, p):
    """t^e mod p by square-and-multiply-by-t"""
    r = 1
    dp = p.bit_length() - 1
    for bit in bin(e)[2:]:
        r = _mod(_square(r), p)
        if bit == "1":
            r <<= 1
            if (r >> dp) & 1:
                r ^= p
    return r


def _div_t(h, p):
    """h / t mod p (p has constant term 1)"""
    return (h >> 1) if not (h & 1) else ((h ^ p) >> 1)


def _to_words(g):
    return np.frombuffer(g.to_bytes(4 * N, "little"), dtype=np.uint32).copy()          # 19968 bits >= degree 19936


def jump_polys(J, levels):
    """[g_k = t^(J * 2^k - 1) mod phi, k = 0 .. levels-1] as (levels, 624) uint32 bit masks (bit b of word w = coefficient of t^(32 w + b))"""
    key = (int(J), int(levels))
    if key not in _CACHE:
        p = phi()
        h = _xpow(J, p)                     # t^J
        out = []
        for _ in range(levels):
            out.append(_to_words(_div_t(h, p)))
            h = _mod(_square(h), p)         # t^(2 J 2^k)
        _CACHE[key] = np.stack(out) if out else np.zeros((0, N), dtype=np.uint32)
    return _CACHE[key]


def _mulmod(a, b, p):
    """a * b mod p over GF(2)"""
    acc = 0
    i = 0
    while b:
        if b & 1:
            acc ^= a << i
        b >>= 1
        i += 1
    return _mod(acc, p)


def jump_polys_from_start(J, nchunks):
    """[t^(c J - 1) mod phi for c = 1 .. nchunks-1] as an (nchunks-1, 624) uint32 array: the masks that take the key at stream position 0 to the keys at positions
    J, 2J, ... in ONE level of independent jumps (cached per (J, nchunks): ~15 ms per polynomial, once)"""
    key = ("start", int(J), int(nchunks))
    if key not in _CACHE:
        p = phi()
        h = _xpow(J, p)
        out, cur = [], h
        for _ in range(1, nchunks):
            out.append(_to_words(_div_t(cur, p)))
            cur = _mulmod(cur, h, p)
        _CACHE[key] = np.stack(out) if out else np.zeros((0, N), dtype=np.uint32)
    return _CACHE[key]


def _stream(key, n):
    """the next n untempered words x_0 .. x_{n-1} of the generator whose key is `key` (x_0 .. x_623 = key)"""
    x = np.empty(max(n, N), dtype=np.uint32)
    x[:N] = key
    UP, LO, A = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)
    k = N
    while k < n:                            # 227 words at a time: word k + j needs words <= k + j - 227
        m = min(227, n - k)
        y = (x[k - N:k - N + m] & UP) | (x[k - N + 1:k - N + 1 + m] & LO)
        x[k:k + m] = x[k - 227:k - 227 + m] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), A, np.uint32(0))
        k += m
    return x[:n]


def jump_key(key, g_words):
    """the key J words ahead of `key`, with g_words = one row of jump_polys(J, ...): x_{J+m} = XOR_{i: g_i} x_{i+1+m}, m = 0..623"""
    g = int.from_bytes(np.asarray(g_words, dtype="<u4").tobytes(), "little")
    deg = g.bit_length() - 1
    x = _stream(np.asarray(key, dtype=np.uint32), deg + 1 + N + 1)
    out = np.zeros(N, dtype=np.uint32)
    for i in range(deg + 1):
        if (g >> i) & 1:
            out ^= x[i + 1:i + 1 + N]
    return out
