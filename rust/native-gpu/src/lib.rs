//! GPU-backed bodies for the reference's `native` module, signatures unchanged
//! (/root/reference/src/native/{pairing,miller_loop,final_exponentiation,fin_exp_helpers}.rs).
//! Every function forwards to libbls381_hip.so through `bls381-hip-sys`; results are the same field
//! elements the reference computes (canonical values, compared bit for bit by the parity tests on the C side).
//!
//! In the reference crate these replace the function bodies; `MyFq12` below mirrors
//! `crate::utils::helpers::MyFq12` so that this file compiles on its own.
use ark_bls12_381::{Fq, Fq12, Fq2, G1Affine, G2Affine};
use bls381_hip_sys as sys;
use sys::{check, ctx, fq12_from_limbs, fq12_to_limbs};

// ---- /root/reference/src/utils/helpers.rs:8-76 ------------------------------------------------------
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct MyFq12 {
    pub coeffs: [Fq; 12],
}
/// MyFq12 index k holds ark's flat coefficient MY_TO_ARK[k] (helpers.rs:39-41).
const MY_TO_ARK: [usize; 12] = [0, 6, 2, 8, 4, 10, 1, 7, 3, 9, 5, 11];
impl From<Fq12> for MyFq12 {
    fn from(f: Fq12) -> Self {
        let flat = [f.c0.c0.c0, f.c0.c0.c1, f.c0.c1.c0, f.c0.c1.c1, f.c0.c2.c0, f.c0.c2.c1,
                    f.c1.c0.c0, f.c1.c0.c1, f.c1.c1.c0, f.c1.c1.c1, f.c1.c2.c0, f.c1.c2.c1];
        MyFq12 { coeffs: core::array::from_fn(|k| flat[MY_TO_ARK[k]]) }
    }
}
impl From<MyFq12> for Fq12 {
    fn from(m: MyFq12) -> Self {
        let mut flat = [m.coeffs[0]; 12];
        for k in 0..12 {
            flat[MY_TO_ARK[k]] = m.coeffs[k];
        }
        let c = |i: usize| Fq2::new(flat[2 * i], flat[2 * i + 1]);
        Fq12::new(ark_bls12_381::Fq6::new(c(0), c(1), c(2)), ark_bls12_381::Fq6::new(c(3), c(4), c(5)))
    }
}

// ---- /root/reference/src/native/miller_loop.rs:12-22, 90-119 ---------------------------------------
pub struct G1Prepared(pub G1Affine);
pub type EllCoeff = (Fq2, Fq2, Fq2);
/// Keeps the reference's public fields and additionally the affine point: the GPU recomputes the line
/// coefficients on the fly, so `ell_coeffs` is only filled for callers that read it (`with_coeffs`).
pub struct G2Prepared {
    pub ell_coeffs: Vec<EllCoeff>,
    pub infinity: bool,
    pub affine: G2Affine,
}
impl G2Prepared {
    pub fn is_zero(&self) -> bool {
        self.infinity
    }
    /// G2Prepared::from with the 68 coefficients materialised (bls381_g2_prepare_batch).
    pub fn with_coeffs(q: G2Affine) -> Self {
        let mut me = Self::from(q);
        if !me.infinity {
            let mut g2 = Vec::with_capacity(24);
            sys::push_g2(&mut g2, &q);
            let mut out = vec![0u64; 68 * 36];
            check(unsafe { sys::bls381_g2_prepare_batch(ctx(), g2.as_ptr(), 1, out.as_mut_ptr()) });
            me.ell_coeffs = (0..68)
                .map(|k| {
                    let c = |j: usize| sys::fq2_from_limbs(&out[36 * k + 12 * j..36 * k + 12 * j + 12]);
                    (c(0), c(1), c(2))
                })
                .collect();
        }
        me
    }
}
impl From<G2Affine> for G2Prepared {
    fn from(q: G2Affine) -> Self {
        G2Prepared { ell_coeffs: vec![], infinity: q.infinity, affine: q }
    }
}
impl From<G1Affine> for G1Prepared {
    fn from(p: G1Affine) -> Self {
        G1Prepared(p)
    }
}

fn collect(
    a: impl IntoIterator<Item = impl Into<G1Prepared>>,
    b: impl IntoIterator<Item = impl Into<G2Prepared>>,
) -> (Vec<u64>, Vec<u64>, Vec<u8>) {
    let ps: Vec<G1Affine> = a.into_iter().map(|p| p.into().0).collect();
    let qs: Vec<(G2Affine, bool)> = b.into_iter().map(|q| { let q: G2Prepared = q.into(); (q.affine, q.infinity) }).collect();
    sys::flatten_pairs(&ps, &qs)
}

/// miller_loop.rs:121-162
pub fn multi_miller_loop_native(
    a: impl IntoIterator<Item = impl Into<G1Prepared>>,
    b: impl IntoIterator<Item = impl Into<G2Prepared>>,
) -> Fq12 {
    let (g1, g2, inf) = collect(a, b);
    let mut out = [0u64; 72];
    check(unsafe { sys::bls381_multi_miller_loop(ctx(), g1.as_ptr(), g2.as_ptr(), inf.as_ptr(), inf.len(), out.as_mut_ptr()) });
    fq12_from_limbs(&out)
}

/// miller_loop.rs:185-194
pub fn ell(f: &mut Fq12, g2_coeffs: EllCoeff, p: G1Affine) {
    let fin = fq12_to_limbs(f);
    let mut co = Vec::with_capacity(36);
    sys::push_fq2(&mut co, &g2_coeffs.0);
    sys::push_fq2(&mut co, &g2_coeffs.1);
    sys::push_fq2(&mut co, &g2_coeffs.2);
    let mut g1 = Vec::with_capacity(12);
    sys::push_g1(&mut g1, &p);
    let mut out = [0u64; 72];
    check(unsafe { sys::bls381_ell_batch(ctx(), fin.as_ptr(), co.as_ptr(), g1.as_ptr(), 1, out.as_mut_ptr()) });
    *f = fq12_from_limbs(&out);
}

// ---- /root/reference/src/native/final_exponentiation.rs ----------------------------------------------
fn unary(f: unsafe extern "C" fn(*mut sys::Bls381Ctx, *const u64, usize, *mut u64) -> i32, x: &Fq12) -> Fq12 {
    let xin = fq12_to_limbs(x);
    let mut out = [0u64; 72];
    check(unsafe { f(ctx(), xin.as_ptr(), 1, out.as_mut_ptr()) });
    fq12_from_limbs(&out)
}
/// final_exponentiation.rs:8-19 (panics on zero like the reference's division)
pub fn easy_part_native(a: MyFq12) -> MyFq12 {
    let xin = fq12_to_limbs(&a.into());
    let mut out = [0u64; 72];
    let mut zero = 0u8;
    check(unsafe { sys::bls381_easy_part_batch(ctx(), xin.as_ptr(), 1, out.as_mut_ptr(), &mut zero) });
    fq12_from_limbs(&out).into()
}
/// final_exponentiation.rs:44-72 (any Fq12: generic squarings)
pub fn hard_part_native(r: Fq12) -> Fq12 {
    unary(sys::bls381_hard_part_batch, &r)
}
/// final_exponentiation.rs:74-78
pub fn final_exponentiation(a: MyFq12) -> Fq12 {
    let xin = fq12_to_limbs(&a.into());
    let mut out = [0u64; 72];
    let mut zero = 0u8;
    check(unsafe { sys::bls381_final_exp_batch(ctx(), xin.as_ptr(), 1, out.as_mut_ptr(), &mut zero) });
    fq12_from_limbs(&out)
}

// ---- /root/reference/src/native/fin_exp_helpers.rs ----------------------------------------------------
/// fin_exp_helpers.rs:10-47
pub fn frobenius_map_native(a: MyFq12, power: usize) -> MyFq12 {
    let xin = fq12_to_limbs(&a.into());
    let mut out = [0u64; 72];
    check(unsafe { sys::bls381_frobenius_batch(ctx(), xin.as_ptr(), (power % 12) as u32, 1, out.as_mut_ptr()) });
    fq12_from_limbs(&out).into()
}
/// fin_exp_helpers.rs:60-82
pub fn pow_native(a: MyFq12, exp: Vec<u64>) -> MyFq12 {
    let xin = fq12_to_limbs(&a.into());
    let mut out = [0u64; 72];
    check(unsafe { sys::bls381_pow_batch(ctx(), xin.as_ptr(), exp.as_ptr(), exp.len(), 1, out.as_mut_ptr()) });
    fq12_from_limbs(&out).into()
}
/// fin_exp_helpers.rs:132-142
pub fn conjugate_fp12(a: MyFq12) -> MyFq12 {
    let f: Fq12 = a.into();
    unary(sys::bls381_conjugate_batch, &f).into()
}

// ---- /root/reference/src/native/pairing.rs:8-13 -------------------------------------------------------
pub fn native_pairing(
    a: impl IntoIterator<Item = impl Into<G1Prepared>>,
    b: impl IntoIterator<Item = impl Into<G2Prepared>>,
) -> Fq12 {
    let (g1, g2, inf) = collect(a, b);
    let mut out = [0u64; 72];
    let mut one = 0i32;
    check(unsafe { sys::bls381_multi_pairing(ctx(), g1.as_ptr(), g2.as_ptr(), inf.as_ptr(), inf.len(), out.as_mut_ptr(), &mut one) });
    fq12_from_limbs(&out)
}

/// Not in the reference: n independent pairings in one call (what the GPU is for).
pub fn native_pairing_batch(ps: &[G1Affine], qs: &[G2Affine]) -> Vec<Fq12> {
    let qs: Vec<(G2Affine, bool)> = qs.iter().map(|q| (*q, q.infinity)).collect();
    let (g1, g2, inf) = sys::flatten_pairs(ps, &qs);
    let mut out = vec![0u64; 72 * ps.len()];
    check(unsafe { sys::bls381_pairing_batch(ctx(), g1.as_ptr(), g2.as_ptr(), inf.as_ptr(), ps.len(), out.as_mut_ptr()) });
    out.chunks(72).map(fq12_from_limbs).collect()
}

#[cfg(test)]
mod tests {
    // the reference's own comparisons (pairing.rs:26-34, miller_loop.rs:204-214, final_exponentiation.rs:90-101)
    use super::*;
    use ark_ec::pairing::Pairing;
    use ark_ff::UniformRand;

    #[test]
    fn test_native_pairing() {
        let rng = &mut ark_std::test_rng();
        let (p0, q0) = (G1Affine::rand(rng), G2Affine::rand(rng));
        assert_eq!(ark_bls12_381::Bls12_381::pairing(p0, q0).0, native_pairing([G1Prepared(p0)], [G2Prepared::from(q0)]));
    }
    #[test]
    fn test_native_multi_miller_loop() {
        let rng = &mut ark_std::test_rng();
        let (p0, q0) = (G1Affine::rand(rng), G2Affine::rand(rng));
        assert_eq!(ark_bls12_381::Bls12_381::miller_loop(p0, q0).0, multi_miller_loop_native([G1Prepared(p0)], [G2Prepared::from(q0)]));
    }
    #[test]
    fn test_native_final_exponentiation() {
        use ark_ec::pairing::MillerLoopOutput;
        let x = Fq12::rand(&mut ark_std::test_rng());
        let want = ark_bls12_381::Bls12_381::final_exponentiation(MillerLoopOutput(x)).unwrap().0;
        assert_eq!(want, final_exponentiation(x.into()));
    }
}
