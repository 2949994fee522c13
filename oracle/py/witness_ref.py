"""CPU oracle for the witness trace of the reference's Plonky2 pairing circuit.

TEST INFRASTRUCTURE ONLY (see bls381_ref.py).  "parity unpinned": the reference
has no fixture holding witness values; this replays, call for call, the builder
methods cited below on concrete values and records the value every non-native
Fq operation would write into its 12-limb result target.

A trace row is (kind, value) with kind in
  'add','sub','mul','neg','inv'  one FqTarget op (/root/reference/src/fields/fq_target.rs:144-187)
  'gen'                          one Fq written by Fq12InverseGenerator
                                 (/root/reference/src/fields/fq12_target.rs:374-397)
Constants (FqTarget::constant) are not witness values and produce no row.
"""
import bls381_ref as R

P = R.P

class Trace:
    def __init__(self):
        self.rows = []          # list of (kind, value)
        self.operands = []      # per row: (a, b) as the gadget received them (b None for neg / inv, both None for 'gen')
        self.counts = {'add': 0, 'sub': 0, 'mul': 0, 'neg': 0, 'inv': 0, 'gen': 0}
    def emit(self, kind, v, a=None, b=None):
        self.rows.append((kind, v))
        self.operands.append((a, b))
        self.counts[kind] += 1
        return v
    def aux(self):
        """The auxiliary values the non-native gadgets' generators compute beside each row's value (the gadgets are
        plonky2_ecdsa's, absent from the reference tree: values by definition, layout unpinned): per mul row the
        quotient (a b - r) / p, per row a carry byte (add: a + b >= p, sub: a < b, neg: a != 0), and the inv row's
        quotient (a a^-1 - 1) / p."""
        mul_quot, carry, inv_quot = [], [], None
        for (kind, r), (a, b) in zip(self.rows, self.operands):
            if kind == 'mul':
                assert (a * b - r) % P == 0
                mul_quot.append((a * b - r) // P)
                carry.append(0)
            elif kind == 'add':
                carry.append(1 if a + b >= P else 0)
            elif kind == 'sub':
                carry.append(1 if a < b else 0)
            elif kind == 'neg':
                carry.append(1 if a != 0 else 0)
            elif kind == 'inv':
                assert (a * r - 1) % P == 0
                inv_quot = (a * r - 1) // P
                carry.append(0)
            else:
                carry.append(0)
        return mul_quot, carry, inv_quot

# ---- FqTarget ops (fq_target.rs:144-187): values are ints
def fq_add(t, a, b): return t.emit('add', (a + b) % P, a, b)
def fq_sub(t, a, b): return t.emit('sub', (a - b) % P, a, b)
def fq_mul(t, a, b): return t.emit('mul', a * b % P, a, b)
def fq_neg(t, a): return t.emit('neg', (-a) % P, a)
def fq_inv(t, a): return t.emit('inv', R.fq_inv(a), a)

# ---- Fq2Target (fq2_target.rs): values are [c0, c1]
def fq2_add(t, a, b): return [fq_add(t, a[0], b[0]), fq_add(t, a[1], b[1])]          # :94-104
def fq2_sub(t, a, b): return [fq_sub(t, a[0], b[0]), fq_sub(t, a[1], b[1])]          # :117-127
def fq2_neg(t, a): return [fq_neg(t, a[0]), fq_neg(t, a[1])]                          # :106-115
def fq2_mul(t, a, b):                                                                 # :145-163
    a0b0 = fq_mul(t, a[0], b[0])
    a1b1 = fq_mul(t, a[1], b[1])
    c0 = fq_sub(t, a0b0, a1b1)
    a0b1 = fq_mul(t, a[0], b[1])
    a1b0 = fq_mul(t, a[1], b[0])
    c1 = fq_add(t, a0b1, a1b0)
    return [c0, c1]
def fq2_simple_square(t, a): return fq2_mul(t, a, a)                                  # :218-220
def fq2_double(t, a): return fq2_add(t, a, a)                                         # :222-224
def fq2_mul_assign_by_fp(t, a, s): return [fq_mul(t, a[0], s), fq_mul(t, a[1], s)]    # :226-234
def fq2_mul_by_nonresidue(t, a): return [fq_sub(t, a[0], a[1]), fq_add(t, a[0], a[1])]  # :236-248
def fq2_conjugate(t, a): return [a[0], fq_neg(t, a[1])]                               # :250-256

# ---- Fq6Target (fq6_target.rs): 6 ints in MyFq6 order [c00,c10,c20,c01,c11,c21]
def fq6_add(t, a, b): return [fq_add(t, x, y) for x, y in zip(a, b)]                  # :96-106
def fq6_sub(t, a, b): return [fq_sub(t, x, y) for x, y in zip(a, b)]                  # :119-129
def _fq6_split(a): return [a[0], a[3]], [a[1], a[4]], [a[2], a[5]]
def _fq6_join(t1, t2, t3): return [t1[0], t2[0], t3[0], t1[1], t2[1], t3[1]]
def fq6_mul_by_01(t, s, c0, c1):                                                      # :198-239
    s0, s1, s2 = _fq6_split(s)
    a_a = fq2_mul(t, s0, c0)
    b_b = fq2_mul(t, s1, c1)
    t1 = fq2_mul(t, s2, c1)
    t1 = fq2_mul_by_nonresidue(t, t1)
    t1 = fq2_add(t, t1, a_a)
    c0_add_c1 = fq2_add(t, c0, c1)
    s0_add_s1 = fq2_add(t, s0, s1)
    t2 = fq2_mul(t, c0_add_c1, s0_add_s1)
    t2 = fq2_sub(t, t2, a_a)
    t2 = fq2_sub(t, t2, b_b)
    t3 = fq2_mul(t, s2, c0)
    t3 = fq2_add(t, t3, b_b)
    return _fq6_join(t1, t2, t3)
def fq6_mul_by_1(t, s, c1):                                                           # :241-279
    s0, s1, s2 = _fq6_split(s)
    b_b = fq2_mul(t, s1, c1)
    tmp = fq2_add(t, s1, s2)
    t1 = fq2_mul(t, c1, tmp)
    t1 = fq2_sub(t, t1, b_b)
    t1 = fq2_mul_by_nonresidue(t, t1)
    tmp = fq2_add(t, s0, s1)
    t2 = fq2_mul(t, c1, tmp)
    t2 = fq2_sub(t, t2, b_b)
    return _fq6_join(t1, t2, b_b)
def fq6_mul_by_nonresidue(t, s):                                                      # :282-311
    s0, s1, s2 = _fq6_split(s)
    c0 = fq2_mul_by_nonresidue(t, s2)
    return _fq6_join(c0, s0, s1)

# ---- Fq12Target (fq12_target.rs): 12 ints in MyFq12 order
def fq12_mul(t, a, b):                                                                # :118-188
    a0b0 = []; a0b1 = []; a1b0 = []; a1b1 = []
    for i in range(6):
        for j in range(6):
            c00 = fq_mul(t, a[i], b[j])
            c01 = fq_mul(t, a[i], b[j + 6])
            c10 = fq_mul(t, a[i + 6], b[j])
            c11 = fq_mul(t, a[i + 6], b[j + 6])
            if i + j < len(a0b0):
                a0b0[i + j] = fq_add(t, a0b0[i + j], c00)
                a0b1[i + j] = fq_add(t, a0b1[i + j], c01)
                a1b0[i + j] = fq_add(t, a1b0[i + j], c10)
                a1b1[i + j] = fq_add(t, a1b1[i + j], c11)
            else:
                a0b0.append(c00); a0b1.append(c01); a1b0.append(c10); a1b1.append(c11)
    m = []; pl = []
    for i in range(11):
        m.append(fq_sub(t, a0b0[i], a1b1[i]))
        pl.append(fq_add(t, a0b1[i], a1b0[i]))
    out = []
    for i in range(6):
        if i < 5:
            term1 = fq_mul(t, m[i + 6], 1)
            term2 = fq_neg(t, pl[i + 6])
            s = fq_add(t, m[i], term1)
            out.append(fq_add(t, s, term2))
        else:
            out.append(m[i])
    for i in range(6):
        if i < 5:
            term2 = fq_mul(t, pl[i + 6], 1)
            s = fq_add(t, pl[i], m[i + 6])
            out.append(fq_add(t, s, term2))
        else:
            out.append(pl[i])
    return out
def _fq12_to_fq6(a):                                                                  # :190-239
    return [a[0], a[2], a[4], a[6], a[8], a[10]], [a[1], a[3], a[5], a[7], a[9], a[11]]
def _fq12_from_fq6(c0, c1):                                                           # :241-256
    return [c0[0], c1[0], c0[1], c1[1], c0[2], c1[2], c0[3], c1[3], c0[4], c1[4], c0[5], c1[5]]
def fq12_mul_by_014(t, f, c0, c1, c4):                                                # :258-295
    s0, s1 = _fq12_to_fq6(f)
    aa = fq6_mul_by_01(t, s0, c0, c1)
    bb = fq6_mul_by_1(t, s1, c4)
    o = fq2_add(t, c1, c4)
    s1 = fq6_add(t, s1, s0)
    s1 = fq6_mul_by_01(t, s1, c0, o)
    s1 = fq6_sub(t, s1, aa)
    s1 = fq6_sub(t, s1, bb)
    s0 = fq6_mul_by_nonresidue(t, bb)
    s0 = fq6_add(t, s0, aa)
    return _fq12_from_fq6(s0, s1)
def fq12_conjugate(t, a):                                                             # :314-323
    a = list(a)
    for k in (1, 3, 5, 7, 9, 11):
        a[k] = fq_neg(t, a[k])
    return a
def fq12_inv(t, a):                                                                   # :302-312
    inv = R.myfq12_from_fq12(R.fq12_inv(R.myfq12_to_fq12(a)))
    for v in inv:
        t.emit('gen', v)
    fq12_mul(t, a, inv)          # x * inv, connected to ONE
    return inv
def fq12_div(t, a, b):                                                                # :297-300
    inv = fq12_inv(t, b)
    return fq12_mul(t, a, inv)

# ---- curves/g2.rs
def g2_double_in_place(t, r, two_inv):                                                # :59-98
    x, y, z = r
    a = fq2_mul(t, x, y)
    a = fq2_mul_assign_by_fp(t, a, two_inv)
    b = fq2_simple_square(t, y)
    c = fq2_simple_square(t, z)
    c_double = fq2_double(t, c)
    c_double = fq2_add(t, c_double, c)
    e = fq2_mul(t, list(R.COEFF_B_G2), c_double)
    e_double = fq2_double(t, e)
    f = fq2_add(t, e_double, e)
    g = fq2_add(t, b, f)
    g = fq2_mul_assign_by_fp(t, g, two_inv)
    y_add_z = fq2_add(t, y, z)
    y_z_squared = fq2_simple_square(t, y_add_z)
    b_add_c = fq2_add(t, b, c)
    h = fq2_sub(t, y_z_squared, b_add_c)
    i = fq2_sub(t, e, b)
    j = fq2_simple_square(t, x)
    e_square = fq2_simple_square(t, e)
    b_sub_f = fq2_sub(t, b, f)
    nx = fq2_mul(t, a, b_sub_f)
    g_square = fq2_simple_square(t, g)
    e_sq_double = fq2_double(t, e_square)
    e_sq_double = fq2_add(t, e_sq_double, e_square)
    ny = fq2_sub(t, g_square, e_sq_double)
    nz = fq2_mul(t, b, h)
    j_double = fq2_double(t, j)
    j_triple = fq2_add(t, j_double, j)
    return (nx, ny, nz), (i, j_triple, fq2_neg(t, h))
def g2_add_in_place(t, r, q):                                                         # :100-129
    x, y, z = r
    qx, qy = q
    qy_z = fq2_mul(t, qy, z)
    theta = fq2_sub(t, y, qy_z)
    qx_z = fq2_mul(t, qx, z)
    lam = fq2_sub(t, x, qx_z)
    c = fq2_simple_square(t, theta)
    d = fq2_simple_square(t, lam)
    e = fq2_mul(t, lam, d)
    f = fq2_mul(t, z, c)
    g = fq2_mul(t, x, d)
    g_double = fq2_double(t, g)
    e_add_f = fq2_add(t, e, f)
    h = fq2_sub(t, e_add_f, g_double)
    nx = fq2_mul(t, lam, h)
    g_sub_h = fq2_sub(t, g, h)
    e_mul_y = fq2_mul(t, e, y)
    th_g_h = fq2_mul(t, theta, g_sub_h)
    ny = fq2_sub(t, th_g_h, e_mul_y)
    nz = fq2_mul(t, z, e)
    lam_qy = fq2_mul(t, lam, qy)
    th_qx = fq2_mul(t, theta, qx)
    j = fq2_sub(t, th_qx, lam_qy)
    return (nx, ny, nz), (j, fq2_neg(t, theta), lam)
def g2_prepared_from(t, q):                                                           # :148-178
    two = fq_add(t, 1, 1)
    two_inv = fq_inv(t, two)
    qa = ([q[0][0], q[0][1]], [q[1][0], q[1][1]])
    r = (qa[0], qa[1], [1, 0])
    coeffs = []
    for bit in R.bls_x_bits_skip1():
        r, c = g2_double_in_place(t, r, two_inv)
        coeffs.append(c)
        if bit:
            r, c = g2_add_in_place(t, r, qa)
            coeffs.append(c)
    return coeffs

# ---- miller_loop.rs
def ell_target(t, f, coeff, p):                                                       # :63-79
    c0, c1, c2 = coeff
    c2 = fq2_mul_assign_by_fp(t, c2, p[1])
    c1 = fq2_mul_assign_by_fp(t, c1, p[0])
    return fq12_mul_by_014(t, f, c0, c1, c2)
def multi_miller_loop(t, p, coeffs):                                                  # :19-61, one pair
    f = R.myfq12_from_fq12(R.FQ12_ONE)
    k = 0
    for bit in R.bls_x_bits_skip1():
        f = fq12_mul(t, f, f)
        f = ell_target(t, f, coeffs[k], p); k += 1
        if bit:
            f = ell_target(t, f, coeffs[k], p); k += 1
    return fq12_conjugate(t, f)

# ---- final_exponentiation_helpers.rs / final_exponentiation.rs
def frobenius_map(t, a, power):                                                       # helpers :14-57
    pw = power % 12
    g = R.frob_coeffs(pw)
    out = []
    for i in range(6):
        fc = R.fq2_pow(g, i)
        x = [a[i], a[i + 6]]
        if pw % 2:
            x = fq2_conjugate(t, x)
        out.append(x if fc == R.FQ2_ONE else fq2_mul(t, x, [fc[0], fc[1]]))
    return [x[0] for x in out] + [x[1] for x in out]
def pow_target(t, a, exp):                                                            # helpers :59-85
    res = list(a)
    for b in bin(exp)[3:]:
        res = fq12_mul(t, res, res)
        if b == '1':
            res = fq12_mul(t, res, a)
    return res
def easy_part(t, a):                                                                  # :12-21
    f1 = fq12_conjugate(t, a)
    f2 = fq12_div(t, f1, a)
    f3 = frobenius_map(t, f2, 2)
    return fq12_mul(t, f3, f2)
def hard_part(t, r):                                                                  # :23-55
    X = R.BLS_X
    y0 = fq12_mul(t, r, r)
    y1 = pow_target(t, r, X)
    y1 = fq12_conjugate(t, y1)
    y2 = fq12_conjugate(t, r)
    y1 = fq12_mul(t, y1, y2)
    y2 = pow_target(t, y1, X)
    y2 = fq12_conjugate(t, y2)
    y1 = fq12_conjugate(t, y1)
    y1 = fq12_mul(t, y1, y2)
    y2 = pow_target(t, y1, X)
    y2 = fq12_conjugate(t, y2)
    y1 = frobenius_map(t, y1, 1)
    y1 = fq12_mul(t, y1, y2)
    r = fq12_mul(t, r, y0)
    y0 = pow_target(t, y1, X)
    y0 = fq12_conjugate(t, y0)
    y2 = pow_target(t, y0, X)
    y2 = fq12_conjugate(t, y2)
    y0 = frobenius_map(t, y1, 2)
    y1 = fq12_conjugate(t, y1)
    y1 = fq12_mul(t, y1, y2)
    y1 = fq12_mul(t, y1, y0)
    return fq12_mul(t, r, y1)
def final_exponentiation(t, a):                                                       # :57-64
    return hard_part(t, easy_part(t, a))

def pairing_witness(p, q):
    """Replay of src/pairing.rs:13-20 preceded by G2PreparedTarget::from, the way the
    reference's test drives it (src/pairing.rs:60-63).  Returns (Trace, MyFq12 result)."""
    t = Trace()
    coeffs = g2_prepared_from(t, q)
    f = multi_miller_loop(t, p, coeffs)
    out = final_exponentiation(t, f)
    return t, out

# Totals for one single-pair pairing circuit (asserted by the tests).
EXPECTED_COUNTS = {'mul': 70656, 'add': 59819, 'sub': 8616, 'neg': 2272, 'inv': 1, 'gen': 12}
