"""Limb-accurate model of the divstep modular inversion the GPU runs (test infrastructure).

Bernstein-Yang "safegcd" division steps (delta variant), in batches of 30 on the low words, on 13
signed 30-bit limbs - the structure of libsecp256k1's modinv32, sized for the 381-bit BLS12-381 base
field.  1,110 division steps (37 batches) cover the proven bound floor((49 * 381 + 57) / 17) = 1,101.
Every intermediate is checked against the width of the register that holds it in the kernel.
"""
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
M30 = (1 << 30) - 1
NL = 13
BATCHES = 37
P30 = [(P >> (30 * i)) & M30 for i in range(NL)]
PINV30 = pow(P, -1, 1 << 30)


def i32(x):
    assert -(1 << 31) <= x < (1 << 31), "value leaves a 32-bit register"
    return x


def i64(x):
    assert -(1 << 63) <= x < (1 << 63), "value leaves a 64-bit register"
    return x


def to_limbs(v):
    assert 0 <= v < 1 << 390
    return [(v >> (30 * i)) & M30 for i in range(NL)]


def value(l):
    return sum(x << (30 * i) for i, x in enumerate(l))


def divsteps_30(delta, f0, g0):
    """30 division steps on the low 32 bits of f and g.  Returns delta and the transition matrix
    (u, v, q, r) with 2^30 [f'; g'] = [[u, v], [q, r]] [f; g]."""
    u, v, q, r = 1, 0, 0, 1
    f, g = f0 & 0xffffffff, g0 & 0xffffffff
    for _ in range(30):
        assert f & 1
        c1 = delta > 0 and (g & 1)
        if c1:
            delta, f, g = -delta, g, (-f) & 0xffffffff
            u, v, q, r = q, r, -u, -v
        if g & 1:
            g = (g + f) & 0xffffffff
            q, r = q + u, r + v
        delta += 1
        g >>= 1
        u, v = i32(u << 1), i32(v << 1)
        i32(q), i32(r)
    return delta, (u, v, q, r)


def update_fg(f, g, t):
    u, v, q, r = t
    cf = i64(u * f[0] + v * g[0])
    cg = i64(q * f[0] + r * g[0])
    assert cf & M30 == 0 and cg & M30 == 0
    cf >>= 30
    cg >>= 30
    for i in range(1, NL):
        cf = i64(cf + u * f[i] + v * g[i])
        cg = i64(cg + q * f[i] + r * g[i])
        f[i - 1] = cf & M30
        g[i - 1] = cg & M30
        cf >>= 30
        cg >>= 30
    f[NL - 1] = i32(cf)
    g[NL - 1] = i32(cg)


def update_de(d, e, t):
    """[d; e] <- t [d; e] / 2^30 mod p, inputs and outputs in (-2p, p)."""
    u, v, q, r = t
    sd = -1 if d[NL - 1] < 0 else 0
    se = -1 if e[NL - 1] < 0 else 0
    md = (u & sd) + (v & se)
    me = (q & sd) + (r & se)
    cd = i64(u * d[0] + v * e[0])
    ce = i64(q * d[0] + r * e[0])
    md -= (PINV30 * (cd & 0xffffffff) + md) & M30
    me -= (PINV30 * (ce & 0xffffffff) + me) & M30
    i32(md), i32(me)
    cd = i64(cd + P30[0] * md)
    ce = i64(ce + P30[0] * me)
    assert cd & M30 == 0 and ce & M30 == 0
    cd >>= 30
    ce >>= 30
    for i in range(1, NL):
        cd = i64(cd + u * d[i] + v * e[i] + P30[i] * md)
        ce = i64(ce + q * d[i] + r * e[i] + P30[i] * me)
        d[i - 1] = cd & M30
        e[i - 1] = ce & M30
        cd >>= 30
        ce >>= 30
    d[NL - 1] = i32(cd)
    e[NL - 1] = i32(ce)
    assert -2 * P < value(d) < P and -2 * P < value(e) < P


def modinv(x):
    """x^-1 mod p for 0 <= x < p (0 for x = 0), limb for limb as the kernel computes it."""
    f, g = to_limbs(P), to_limbs(x)
    d, e = [0] * NL, [0] * NL
    e[0] = 1
    delta = 1
    for _ in range(BATCHES):
        delta, t = divsteps_30(delta, f[0] | ((f[1] & 3) << 30), g[0] | ((g[1] & 3) << 30))
        update_de(d, e, t)
        update_fg(f, g, t)
    assert value(g) == 0
    fv = value(f)
    assert fv in (1, -1) or x == 0
    r = value(d)
    if fv < 0:
        r = -r
    # (-2p, 2p) -> [0, p)
    if r < 0:
        r += P
    if r < 0:
        r += P
    if r >= P:
        r -= P
    assert 0 <= r < P
    return r


if __name__ == "__main__":
    import random
    rng = random.Random(7)
    xs = [0, 1, 2, P - 1, P - 2, (P + 1) // 2, 1 << 380, (1 << 381) - 1 - (1 << 381) % 1] + [rng.randrange(P) for _ in range(2000)]
    for x in xs:
        x %= P
        got = modinv(x)
        want = pow(x, -1, P) if x else 0
        assert got == want, hex(x)
    print("ok", len(xs))
