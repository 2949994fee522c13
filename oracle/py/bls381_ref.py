"""CPU oracle (big-integer restatement) of the reference's BLS12-381 pairing path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library imports this module;
it is the checker for tests/, __graft_entry__.smoke() and bench.py's CPU
baseline leg.

PARITY STATUS: "parity unpinned" against arkworks.  The reference holds no
golden vectors (its tests compare against ark-bls12-381 0.4.0 at run time on
unseeded random inputs) and no Rust toolchain exists here, so this oracle is
pinned by (a) formula-faithful restatement of the reference source cited per
function, (b) implementation-independent identities: bilinearity, e^r = 1,
final_exp(x) = x^(3(p^12-1)/r), easy(x) = x^((p^6-1)(p^2+1)).

Field tower (ark-bls12-381, used at /root/reference/src/native/miller_loop.rs:1):
  Fq2  = Fq[u]/(u^2+1)           /root/reference/src/fields/fq2_target.rs:236-240
  Fq6  = Fq2[v]/(v^3-(1+u))      /root/reference/src/fields/fq6_target.rs:282-286
  Fq12 = Fq6[w]/(w^2-v)
Values are plain Python ints / nested tuples:
  Fq2 = (c0, c1); Fq6 = (Fq2, Fq2, Fq2); Fq12 = (Fq6, Fq6).
"""

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
# /root/reference/src/utils/constants.rs:1-2
BLS_X = 15132376222941642752
BLS_X_IS_NEGATIVE = True

G1_GEN = (
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
)
G2_GEN = (
    (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
     0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
    (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
     0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
)

# ---------------------------------------------------------------- Fq
def fq_inv(a):
    if a % P == 0:
        raise ZeroDivisionError("inverse of zero in Fq")
    return pow(a, P - 2, P)

# ---------------------------------------------------------------- Fq2
FQ2_ZERO = (0, 0)
FQ2_ONE = (1, 0)

def fq2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def fq2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def fq2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def fq2_double(a): return fq2_add(a, a)
def fq2_conj(a): return (a[0], (-a[1]) % P)
def fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def fq2_square(a): return fq2_mul(a, a)
def fq2_mul_by_fp(a, s): return (a[0] * s % P, a[1] * s % P)
def fq2_mul_by_nonresidue(a):
    # (a + bu)(1 + u) = (a - b) + (a + b)u
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)
def fq2_inv(a):
    n = fq_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * n % P, (-a[1]) * n % P)
def fq2_pow(a, e):
    r = FQ2_ONE
    while e:
        if e & 1:
            r = fq2_mul(r, a)
        a = fq2_mul(a, a)
        e >>= 1
    return r

# ---------------------------------------------------------------- Fq6
FQ6_ZERO = (FQ2_ZERO, FQ2_ZERO, FQ2_ZERO)
FQ6_ONE = (FQ2_ONE, FQ2_ZERO, FQ2_ZERO)

def fq6_add(a, b): return tuple(fq2_add(x, y) for x, y in zip(a, b))
def fq6_sub(a, b): return tuple(fq2_sub(x, y) for x, y in zip(a, b))
def fq6_neg(a): return tuple(fq2_neg(x) for x in a)
def fq6_mul_by_nonresidue(a):
    # (a0 + a1 v + a2 v^2) v = xi*a2 + a0 v + a1 v^2
    return (fq2_mul_by_nonresidue(a[2]), a[0], a[1])
def fq6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = fq2_add(fq2_mul(a0, b0), fq2_mul_by_nonresidue(fq2_add(fq2_mul(a1, b2), fq2_mul(a2, b1))))
    t1 = fq2_add(fq2_add(fq2_mul(a0, b1), fq2_mul(a1, b0)), fq2_mul_by_nonresidue(fq2_mul(a2, b2)))
    t2 = fq2_add(fq2_add(fq2_mul(a0, b2), fq2_mul(a1, b1)), fq2_mul(a2, b0))
    return (t0, t1, t2)
def fq6_inv(a):
    a0, a1, a2 = a
    t0 = fq2_sub(fq2_square(a0), fq2_mul_by_nonresidue(fq2_mul(a1, a2)))
    t1 = fq2_sub(fq2_mul_by_nonresidue(fq2_square(a2)), fq2_mul(a0, a1))
    t2 = fq2_sub(fq2_square(a1), fq2_mul(a0, a2))
    n = fq2_add(fq2_mul(a0, t0), fq2_mul_by_nonresidue(fq2_add(fq2_mul(a2, t1), fq2_mul(a1, t2))))
    ni = fq2_inv(n)
    return (fq2_mul(t0, ni), fq2_mul(t1, ni), fq2_mul(t2, ni))

# ---------------------------------------------------------------- Fq12
FQ12_ONE = (FQ6_ONE, FQ6_ZERO)
FQ12_ZERO = (FQ6_ZERO, FQ6_ZERO)

def fq12_add(a, b): return (fq6_add(a[0], b[0]), fq6_add(a[1], b[1]))
def fq12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    c0 = fq6_add(fq6_mul(a0, b0), fq6_mul_by_nonresidue(fq6_mul(a1, b1)))
    c1 = fq6_add(fq6_mul(a0, b1), fq6_mul(a1, b0))
    return (c0, c1)
def fq12_square(a): return fq12_mul(a, a)
def fq12_conj(a): return (a[0], fq6_neg(a[1]))
def fq12_inv(a):
    a0, a1 = a
    n = fq6_sub(fq6_mul(a0, a0), fq6_mul_by_nonresidue(fq6_mul(a1, a1)))
    if n == FQ6_ZERO:
        raise ZeroDivisionError("inverse of zero in Fq12")
    ni = fq6_inv(n)
    return (fq6_mul(a0, ni), fq6_neg(fq6_mul(a1, ni)))
def fq12_pow(a, e):
    r = FQ12_ONE
    while e:
        if e & 1:
            r = fq12_mul(r, a)
        a = fq12_mul(a, a)
        e >>= 1
    return r
def fq12_mul_by_014(f, c0, c1, c4):
    """ark Fp12::mul_by_014 semantics (restated in-repo at
    /root/reference/src/fields/fq12_target.rs:258-295): f * ((c0,c1,0),(0,c4,0))."""
    return fq12_mul(f, ((c0, c1, FQ2_ZERO), (FQ2_ZERO, c4, FQ2_ZERO)))

def fq12_flatten_ark(a):
    """ark tower order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 (12 ints)."""
    return [c for h in a for q in h for c in q]
def fq12_from_flat_ark(v):
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))

# ------------------------------------------------ MyFq12 (w-power order)
# /root/reference/src/utils/helpers.rs:14-76
_ARK_TO_MY = [0, 6, 2, 8, 4, 10, 1, 7, 3, 9, 5, 11]  # my[k] = ark_flat[_ARK_TO_MY[k]]

def myfq12_from_fq12(a):
    f = fq12_flatten_ark(a)
    return [f[i] for i in _ARK_TO_MY]
def myfq12_to_fq12(m):
    f = [0] * 12
    for k, i in enumerate(_ARK_TO_MY):
        f[i] = m[k]
    return fq12_from_flat_ark(f)
def myfq12_mul(a, b):
    """/root/reference/src/utils/helpers.rs:90-152 (schoolbook, w^6 = 1+u)."""
    a0b0 = [0] * 11; a0b1 = [0] * 11; a1b0 = [0] * 11; a1b1 = [0] * 11
    for i in range(6):
        for j in range(6):
            a0b0[i + j] = (a0b0[i + j] + a[i] * b[j]) % P
            a0b1[i + j] = (a0b1[i + j] + a[i] * b[j + 6]) % P
            a1b0[i + j] = (a1b0[i + j] + a[i + 6] * b[j]) % P
            a1b1[i + j] = (a1b1[i + j] + a[i + 6] * b[j + 6]) % P
    m = [(a0b0[i] - a1b1[i]) % P for i in range(11)]
    pl = [(a0b1[i] + a1b0[i]) % P for i in range(11)]
    out = []
    for i in range(6):
        out.append((m[i] + m[i + 6] - pl[i + 6]) % P if i < 5 else m[i])
    for i in range(6):
        out.append((pl[i] + m[i + 6] + pl[i + 6]) % P if i < 5 else pl[i])
    return out
def myfq12_conj(a):
    """/root/reference/src/native/fin_exp_helpers.rs:132-142."""
    return [c if i % 2 == 0 else (-c) % P for i, c in enumerate(a)]

# ------------------------------------------------ curve helpers (test input generation)
def g1_add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        l = 3 * x1 * x1 * fq_inv(2 * y1) % P
    else:
        l = (y2 - y1) * fq_inv(x2 - x1) % P
    x3 = (l * l - x1 - x2) % P
    return (x3, (l * (x1 - x3) - y1) % P)
def g1_mul(p, k):
    r = None
    while k:
        if k & 1: r = g1_add(r, p)
        p = g1_add(p, p)
        k >>= 1
    return r
def g1_neg(p): return None if p is None else (p[0], (-p[1]) % P)
def g2_add(p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    x1, y1 = p1; x2, y2 = p2
    if x1 == x2:
        if fq2_add(y1, y2) == FQ2_ZERO: return None
        l = fq2_mul(fq2_mul_by_fp(fq2_square(x1), 3), fq2_inv(fq2_double(y1)))
    else:
        l = fq2_mul(fq2_sub(y2, y1), fq2_inv(fq2_sub(x2, x1)))
    x3 = fq2_sub(fq2_sub(fq2_square(l), x1), x2)
    return (x3, fq2_sub(fq2_mul(l, fq2_sub(x1, x3)), y1))
def g2_mul(p, k):
    r = None
    while k:
        if k & 1: r = g2_add(r, p)
        p = g2_add(p, p)
        k >>= 1
    return r
def g2_neg(p): return None if p is None else (p[0], fq2_neg(p[1]))
def g1_on_curve(p): return p is None or (p[1] * p[1] - p[0] ** 3 - 4) % P == 0
def g2_on_curve(p):
    if p is None: return True
    x, y = p
    return fq2_sub(fq2_square(y), fq2_add(fq2_mul(fq2_square(x), x), (4, 4))) == FQ2_ZERO

# ------------------------------------------------ Miller loop
TWO_INV = fq_inv(2)            # /root/reference/src/native/miller_loop.rs:92
COEFF_B_G2 = (4, 4)            # ark g2::Config::COEFF_B, used at miller_loop.rs:55

def g2_double_in_place(r):
    """/root/reference/src/native/miller_loop.rs:50-69. r = (x,y,z) -> (r', coeff)."""
    x, y, z = r
    a = fq2_mul_by_fp(fq2_mul(x, y), TWO_INV)
    b = fq2_square(y)
    c = fq2_square(z)
    e = fq2_mul(COEFF_B_G2, fq2_add(fq2_double(c), c))
    f = fq2_add(fq2_double(e), e)
    g = fq2_mul_by_fp(fq2_add(b, f), TWO_INV)
    h = fq2_sub(fq2_square(fq2_add(y, z)), fq2_add(b, c))
    i = fq2_sub(e, b)
    j = fq2_square(x)
    e_square = fq2_square(e)
    nx = fq2_mul(a, fq2_sub(b, f))
    ny = fq2_sub(fq2_square(g), fq2_add(fq2_double(e_square), e_square))
    nz = fq2_mul(b, h)
    return (nx, ny, nz), (i, fq2_add(fq2_double(j), j), fq2_neg(h))

def g2_add_in_place(r, q):
    """/root/reference/src/native/miller_loop.rs:71-87."""
    x, y, z = r
    qx, qy = q
    theta = fq2_sub(y, fq2_mul(qy, z))
    lam = fq2_sub(x, fq2_mul(qx, z))
    c = fq2_square(theta)
    d = fq2_square(lam)
    e = fq2_mul(lam, d)
    f = fq2_mul(z, c)
    g = fq2_mul(x, d)
    h = fq2_sub(fq2_add(e, f), fq2_double(g))
    nx = fq2_mul(lam, h)
    ny = fq2_sub(fq2_mul(theta, fq2_sub(g, h)), fq2_mul(e, y))
    nz = fq2_mul(z, e)
    j = fq2_sub(fq2_mul(theta, qx), fq2_mul(lam, qy))
    return (nx, ny, nz), (j, fq2_neg(theta), lam)

def bls_x_bits_skip1():
    """BitIteratorBE::new([BLS_X]).skip(1): BLS_X has its bit 63 set, so this is bits 62..0."""
    return [(BLS_X >> i) & 1 for i in range(62, -1, -1)]

def g2_prepare(q):
    """/root/reference/src/native/miller_loop.rs:90-119. q=None is infinity -> []."""
    if q is None:
        return []
    r = (q[0], q[1], FQ2_ONE)
    coeffs = []
    for bit in bls_x_bits_skip1():
        r, c = g2_double_in_place(r)
        coeffs.append(c)
        if bit:
            r, c = g2_add_in_place(r, q)
            coeffs.append(c)
    return coeffs

def ell(f, coeff, p):
    """/root/reference/src/native/miller_loop.rs:185-194."""
    c0, c1, c2 = coeff
    px, py = p
    c2 = fq2_mul_by_fp(c2, py)
    c1 = fq2_mul_by_fp(c1, px)
    return fq12_mul_by_014(f, c0, c1, c2)

def multi_miller_loop(ps, qs):
    """/root/reference/src/native/miller_loop.rs:121-162 (chunks of 4 share the squarings)."""
    if len(ps) != len(qs):
        raise ValueError("zip_eq: length mismatch")
    pairs = [(p, g2_prepare(q)) for p, q in zip(ps, qs) if p is not None and q is not None]
    f_total = FQ12_ONE
    for s in range(0, len(pairs), 4):
        chunk = pairs[s:s + 4]
        idx = [0] * len(chunk)
        f = FQ12_ONE
        for bit in bls_x_bits_skip1():
            f = fq12_square(f)
            for k, (p, cs) in enumerate(chunk):
                f = ell(f, cs[idx[k]], p); idx[k] += 1
            if bit:
                for k, (p, cs) in enumerate(chunk):
                    f = ell(f, cs[idx[k]], p); idx[k] += 1
        f_total = fq12_mul(f_total, f)
    if BLS_X_IS_NEGATIVE:
        f_total = fq12_conj(f_total)
    return f_total

# ------------------------------------------------ final exponentiation
def frob_coeffs(index):
    """/root/reference/src/native/fin_exp_helpers.rs:49-58."""
    return fq2_pow((1, 1), (P ** index - 1) // 6)

def frobenius_map_native(a, power):
    """/root/reference/src/native/fin_exp_helpers.rs:10-47 on a MyFq12 (list of 12)."""
    pw = power % 12
    g = frob_coeffs(pw)
    out = []
    for i in range(6):
        fc = fq2_pow(g, i)
        x = (a[i], a[i + 6])
        if pw % 2:
            x = fq2_conj(x)
        out.append(x if fc == FQ2_ONE else fq2_mul(x, fc))
    return [x[0] for x in out] + [x[1] for x in out]

def pow_native(a, exp):
    """/root/reference/src/native/fin_exp_helpers.rs:60-130: get_naf never carries, so
    this is plain MSB-first square-and-multiply by the bits of exp (a MyFq12 list)."""
    bits = bin(exp)[2:]
    res = list(a)
    for b in bits[1:]:
        res = myfq12_mul(res, res)
        if b == '1':
            res = myfq12_mul(res, a)
    return res

def easy_part_native(a):
    """/root/reference/src/native/final_exponentiation.rs:8-19 (MyFq12 in/out)."""
    f1 = myfq12_conj(a)
    f2 = myfq12_from_fq12(fq12_mul(myfq12_to_fq12(f1), fq12_inv(myfq12_to_fq12(a))))
    f3 = frobenius_map_native(f2, 2)
    return myfq12_mul(f3, f2)

def hard_part_native(r):
    """/root/reference/src/native/final_exponentiation.rs:44-72 (Fq12 in/out)."""
    M = myfq12_from_fq12; F = myfq12_to_fq12
    y0 = fq12_square(r)
    y1 = fq12_conj(F(pow_native(M(r), BLS_X)))
    y2 = fq12_conj(r)
    y1 = fq12_mul(y1, y2)
    y2 = fq12_conj(F(pow_native(M(y1), BLS_X)))
    y1 = fq12_conj(y1)
    y1 = fq12_mul(y1, y2)
    y2 = fq12_conj(F(pow_native(M(y1), BLS_X)))
    y1 = F(frobenius_map_native(M(y1), 1))
    y1 = fq12_mul(y1, y2)
    r = fq12_mul(r, y0)
    y0 = fq12_conj(F(pow_native(M(y1), BLS_X)))
    y2 = fq12_conj(F(pow_native(M(y0), BLS_X)))
    y0 = F(frobenius_map_native(M(y1), 2))
    y1 = fq12_conj(y1)
    y1 = fq12_mul(y1, y2)
    y1 = fq12_mul(y1, y0)
    return fq12_mul(r, y1)

def final_exponentiation(a):
    """/root/reference/src/native/final_exponentiation.rs:74-78 (Fq12 in/out; the reference
    takes a MyFq12 — the permutation is applied here)."""
    return hard_part_native(myfq12_to_fq12(easy_part_native(myfq12_from_fq12(a))))

FINAL_EXP_EXPONENT = 3 * (P ** 12 - 1) // R_ORDER
EASY_EXPONENT = (P ** 6 - 1) * (P ** 2 + 1)

def final_exponentiation_naive(a):
    """Known answer: the reference's chain computes x^(3 (p^12-1)/r)."""
    return fq12_pow(a, FINAL_EXP_EXPONENT)

def native_pairing(ps, qs):
    """/root/reference/src/native/pairing.rs:8-13."""
    return final_exponentiation(multi_miller_loop(ps, qs))

# ------------------------------------------------ limb / byte formats
MONT_R = (1 << 384) % P

def to_mont(a): return a * MONT_R % P
def from_mont(a): return a * fq_inv(MONT_R) % P
def fq_to_u64_limbs(a): return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]
def fq_to_u32_limbs(a): return [(a >> (32 * i)) & 0xFFFFFFFF for i in range(12)]
def u32_limbs_to_int(l): return sum(int(x) << (32 * i) for i, x in enumerate(l))
def u64_limbs_to_int(l): return sum(int(x) << (64 * i) for i, x in enumerate(l))


# ---------------------------------------------------------------- the other witness generators
# Values the in-repo generators of the reference put into the witness; each output is determined by
# its inputs alone (the square roots through the requested sign), so they are pinned by mathematics:
# out^2 = x, sgn0(out) = sgn, out * x = 1.
def sgn0_fq(x):
    """/root/reference/src/utils/helpers.rs:159-167: parity of the canonical integer."""
    return (x % P) & 1 == 1

def sgn0_fq2(x):
    """/root/reference/src/utils/helpers.rs:169-174."""
    return sgn0_fq(x[0]) or (x[0] % P == 0 and sgn0_fq(x[1]))

def fq_sqrt(x):
    """A square root of x in Fq (p = 3 mod 4), or None."""
    r = pow(x % P, (P + 1) // 4, P)
    return r if r * r % P == x % P else None

def fq2_sqrt(a):
    """A square root of a in Fq2 = Fq[u]/(u^2+1), or None."""
    a0, a1 = a[0] % P, a[1] % P
    if a1 == 0:
        s = fq_sqrt(a0)
        if s is not None:
            return (s, 0)
        return (0, fq_sqrt((-a0) % P))          # -1 is a non-residue: -a0 is a square
    g = fq_sqrt((a0 * a0 + a1 * a1) % P)
    if g is None:
        return None
    half = (P + 1) // 2
    x0 = fq_sqrt((a0 + g) * half % P)
    if x0 is None:
        x0 = fq_sqrt((a0 - g) * half % P)
    x1 = a1 * fq_inv(2 * x0 % P) % P
    r = (x0, x1)
    assert fq2_square(r) == (a0, a1)
    return r

def fq_sqrt_generator(x, sgn):
    """FqSqrtGenerator::run_once, /root/reference/src/fields/fq_target.rs:296-323: the square root
    whose parity bit equals sgn.  Raises where the reference panics (no root, or root 0 with sgn set)."""
    r = fq_sqrt(x)
    if r is None:
        raise ValueError("sqrt of a non-residue")
    if sgn0_fq(r) != bool(sgn):
        r = (-r) % P
    if sgn0_fq(r) != bool(sgn):
        raise ValueError("requested sign cannot be met")
    return r

def fq2_sqrt_generator(x, sgn):
    """Fq2SqrtGenerator::run_once, /root/reference/src/fields/fq2_target.rs:401-428."""
    r = fq2_sqrt(x)
    if r is None:
        raise ValueError("sqrt of a non-square")
    if sgn0_fq2(r) != bool(sgn):
        r = fq2_neg(r)
    if sgn0_fq2(r) != bool(sgn):
        raise ValueError("requested sign cannot be met")
    return r

def fq2_inverse_generator(x):
    """Fq2InverseGenerator::run_once, /root/reference/src/fields/fq2_target.rs:347-369: zero maps to zero."""
    if x[0] % P == 0 and x[1] % P == 0:
        return (0, 0)
    return fq2_inv(x)

def fq6_inverse_generator(coeffs):
    """Fq6InverseGenerator::run_once, /root/reference/src/fields/fq6_target.rs:350-371, on MyFq6 order
    [c0.c0, c1.c0, c2.c0, c0.c1, c1.c1, c2.c1] (/root/reference/src/utils/my_fq6.rs:12-44).  Raises on zero."""
    c = [v % P for v in coeffs]
    a = ((c[0], c[3]), (c[1], c[4]), (c[2], c[5]))
    if all(v == 0 for v in c):
        raise ZeroDivisionError("inverse of zero in Fq6")
    r = fq6_inv(a)
    return [r[0][0], r[1][0], r[2][0], r[0][1], r[1][1], r[2][1]]
