//! FFI to libbls381_hip.so (include/bls381_hip.h) and the arkworks <-> limb-buffer conversions.
//!
//! Layouts (include/bls381_hip.h): Fq = 6 x u64 Montgomery limbs (ark_ff's in-memory BigInt<6>),
//! G1 = {x, y}, G2 = {x.c0, x.c1, y.c0, y.c1}, Fq12 = 12 Fq in ark tower order
//! c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1.
use std::ffi::CStr;
use std::os::raw::c_char;
use std::sync::OnceLock;

use ark_bls12_381::{Fq, Fq12, Fq2, Fq6, G1Affine, G2Affine};
use ark_ff::BigInt;

#[repr(C)]
pub struct Bls381Ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct G2PreparedDev {
    _p: [u8; 0],
}

pub const BLS381_OK: i32 = 0;
pub const BLS381_ERR_INVALID: i32 = -1;
pub const BLS381_ERR_NO_DEVICE: i32 = -2;
pub const BLS381_ERR_HIP: i32 = -3;
pub const BLS381_ERR_ZERO: i32 = -4;
pub const BLS381_ERR_POINT: i32 = -5;
pub const BLS381_NOALLOC_MAX_PAIRS: usize = 4096;

extern "C" {
    pub fn bls381_ctx_create(out: *mut *mut Bls381Ctx, device_id: i32) -> i32;
    pub fn bls381_ctx_create_ex(out: *mut *mut Bls381Ctx, device_id: i32, lanes_per_group: i32, lds_slots: i32) -> i32;
    pub fn bls381_ctx_destroy(ctx: *mut Bls381Ctx);
    pub fn bls381_last_error() -> *const c_char;

    pub fn bls381_miller_loop_batch(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_final_exp_batch(ctx: *mut Bls381Ctx, x: *const u64, n: usize, out: *mut u64, zero: *mut u8) -> i32;
    pub fn bls381_pairing_batch(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_multi_miller_loop(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_multi_pairing(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64, is_one: *mut i32) -> i32;
    pub fn bls381_finish_multi_pairing(ctx: *mut Bls381Ctx, partials: *const u64, n: usize, out: *mut u64, is_one: *mut i32) -> i32;
    pub fn bls381_g2_prepare_batch(ctx: *mut Bls381Ctx, g2: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bls381_ell_batch(ctx: *mut Bls381Ctx, f: *const u64, coeffs: *const u64, g1: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bls381_easy_part_batch(ctx: *mut Bls381Ctx, x: *const u64, n: usize, out: *mut u64, zero: *mut u8) -> i32;
    pub fn bls381_hard_part_batch(ctx: *mut Bls381Ctx, x: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bls381_frobenius_batch(ctx: *mut Bls381Ctx, x: *const u64, power: u32, n: usize, out: *mut u64) -> i32;
    pub fn bls381_conjugate_batch(ctx: *mut Bls381Ctx, x: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bls381_pow_batch(ctx: *mut Bls381Ctx, x: *const u64, exp: *const u64, exp_words: usize, n: usize, out: *mut u64) -> i32;
    pub fn bls381_pow_x_batch(ctx: *mut Bls381Ctx, x: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bls381_fq12_mul_batch(ctx: *mut Bls381Ctx, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> i32;

    pub fn bls381_fq12_inverse_limbs_batch(ctx: *mut Bls381Ctx, x: *const u32, n: usize, inv: *mut u32, err: *mut u8) -> i32;
    pub fn bls381_pairing_witness_rows() -> usize;
    pub fn bls381_pairing_witness_kinds(kinds: *mut u8, capacity: usize) -> i32;
    pub fn bls381_pairing_witness_batch(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, n: usize, trace: *mut u32) -> i32;
    pub fn bls381_pairing_witness_batch_dev(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, n: usize, trace: *mut u32) -> i32;
    pub fn bls381_pairing_witness_mul_rows() -> usize;
    pub fn bls381_pairing_witness_aux_batch(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, n: usize, trace: *mut u32, mul_quot: *mut u32,
                                            carry: *mut u8, inv_quot: *mut u32) -> i32;

    pub fn bls381_pairing_batch_dev(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_pairing_batch_dev_noalloc(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_miller_loop_batch_dev(ctx: *mut Bls381Ctx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_final_exp_batch_dev(ctx: *mut Bls381Ctx, x: *const u64, n: usize, out: *mut u64, zero: *mut u8) -> i32;

    pub fn bls381_g2_prepared_create(ctx: *mut Bls381Ctx, g2: *const u64, q_is_infinity: i32, out: *mut *mut G2PreparedDev) -> i32;
    pub fn bls381_g2_prepared_destroy(h: *mut G2PreparedDev);
    pub fn bls381_miller_loop_fixed_q_batch(ctx: *mut Bls381Ctx, q: *const G2PreparedDev, g1: *const u64, p_inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_pairing_fixed_q_batch(ctx: *mut Bls381Ctx, q: *const G2PreparedDev, g1: *const u64, p_inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_pairing_fixed_q_batch_dev(ctx: *mut Bls381Ctx, q: *const G2PreparedDev, g1: *const u64, p_inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_ctx_reserve(ctx: *mut Bls381Ctx, max_items: usize) -> i32;

    pub fn bls381_miller_loop_two_pairs_batch(ctx: *mut Bls381Ctx, g1a: *const u64, g2a: *const u64, g1b: *const u64, g2b: *const u64,
                                              inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_pairing_two_pairs_batch(ctx: *mut Bls381Ctx, g1a: *const u64, g2a: *const u64, g1b: *const u64, g2b: *const u64,
                                          inf: *const u8, n: usize, out: *mut u64, is_one: *mut u8) -> i32;
    pub fn bls381_pairing_two_pairs_batch_dev(ctx: *mut Bls381Ctx, g1a: *const u64, g2a: *const u64, g1b: *const u64, g2b: *const u64,
                                              inf: *const u8, n: usize, out: *mut u64, is_one: *mut u8) -> i32;
    pub fn bls381_verify_batch(ctx: *mut Bls381Ctx, pk48: *const u8, msg96: *const u8, sig96: *const u8, n: usize, check_subgroup_mask: i32,
                               verdict: *mut u8, status: *mut u8) -> i32;
    pub fn bls381_fast_aggregate_verify(ctx: *mut Bls381Ctx, pk48: *const u8, n_keys: usize, msg96: *const u8, sig96: *const u8,
                                        check_subgroup_mask: i32, valid: *mut i32, key_status: *mut u8) -> i32;
    pub fn bls381_fast_aggregate_verify_batch(ctx: *mut Bls381Ctx, pk48: *const u8, n_checks: usize, keys_per_check: usize, msg96: *const u8,
                                              sig96: *const u8, check_subgroup_mask: i32, verdict: *mut u8, key_status: *mut u8) -> i32;

    pub fn bls381_g1_decompress_batch(ctx: *mut Bls381Ctx, in48: *const u8, n: usize, check_subgroup: i32, g1: *mut u64, status: *mut u8) -> i32;
    pub fn bls381_g2_decompress_batch(ctx: *mut Bls381Ctx, in96: *const u8, n: usize, check_subgroup: i32, g2: *mut u64, status: *mut u8) -> i32;
    pub fn bls381_g1_compress_batch(ctx: *mut Bls381Ctx, g1: *const u64, inf: *const u8, n: usize, out48: *mut u8) -> i32;
    pub fn bls381_g2_compress_batch(ctx: *mut Bls381Ctx, g2: *const u64, inf: *const u8, n: usize, out96: *mut u8) -> i32;
    pub fn bls381_g1_aggregate_batch(ctx: *mut Bls381Ctx, g1: *const u64, inf: *const u8, n_groups: usize, group_size: usize, out: *mut u64, out_inf: *mut u8) -> i32;
    pub fn bls381_g2_aggregate_batch(ctx: *mut Bls381Ctx, g2: *const u64, inf: *const u8, n_groups: usize, group_size: usize, out: *mut u64, out_inf: *mut u8) -> i32;

    pub fn bls381_fq_sqrt_limbs_batch(ctx: *mut Bls381Ctx, x: *const u32, sgn: *const u8, n: usize, out: *mut u32, err: *mut u8) -> i32;
    pub fn bls381_fq2_sqrt_limbs_batch(ctx: *mut Bls381Ctx, x: *const u32, sgn: *const u8, n: usize, out: *mut u32, err: *mut u8) -> i32;
    pub fn bls381_fq2_inverse_limbs_batch(ctx: *mut Bls381Ctx, x: *const u32, n: usize, out: *mut u32, err: *mut u8) -> i32;
    pub fn bls381_fq6_inverse_limbs_batch(ctx: *mut Bls381Ctx, x: *const u32, n: usize, out: *mut u32, err: *mut u8) -> i32;

    // several GPUs behind one handle (contiguous shards, one host thread per device)
    pub fn bls381_multi_ctx_create(out: *mut *mut Bls381MultiCtx, device_ids: *const i32, n_devices: i32) -> i32;
    pub fn bls381_multi_ctx_destroy(m: *mut Bls381MultiCtx);
    pub fn bls381_multi_ctx_pairing_batch(m: *mut Bls381MultiCtx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_multi_ctx_miller_loop_batch(m: *mut Bls381MultiCtx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_multi_ctx_devices(m: *const Bls381MultiCtx) -> i32;
    pub fn bls381_multi_ctx_final_exp_batch(m: *mut Bls381MultiCtx, x: *const u64, n: usize, out: *mut u64, zero: *mut u8) -> i32;
    pub fn bls381_multi_ctx_multi_miller_loop(m: *mut Bls381MultiCtx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64) -> i32;
    pub fn bls381_multi_ctx_multi_pairing(m: *mut Bls381MultiCtx, g1: *const u64, g2: *const u64, inf: *const u8, n: usize, out: *mut u64, is_one: *mut i32) -> i32;
    pub fn bls381_multi_ctx_pairing_witness_batch(m: *mut Bls381MultiCtx, g1: *const u64, g2: *const u64, n: usize, trace: *mut u32) -> i32;
    pub fn bls381_multi_ctx_verify_batch(m: *mut Bls381MultiCtx, pk48: *const u8, msg96: *const u8, sig96: *const u8, n: usize,
                                         check_subgroup_mask: i32, verdict: *mut u8, status: *mut u8) -> i32;
}

#[repr(C)]
pub struct Bls381MultiCtx {
    _p: [u8; 0],
}

/// The reference panics where the C ABI returns an error (zip_eq length mismatch, division by zero);
/// the drop-in functions keep that behaviour.
pub fn check(rc: i32) {
    if rc != BLS381_OK {
        let msg = unsafe { CStr::from_ptr(bls381_last_error()) }.to_string_lossy().into_owned();
        panic!("bls381_hip error {rc}: {msg}");
    }
}

struct CtxHandle(*mut Bls381Ctx);
// the library serialises calls on one context internally
unsafe impl Send for CtxHandle {}
unsafe impl Sync for CtxHandle {}

/// One context per process on device BLS381_HIP_DEVICE (default 0), created on first use.
pub fn ctx() -> *mut Bls381Ctx {
    static CTX: OnceLock<CtxHandle> = OnceLock::new();
    CTX.get_or_init(|| {
        let dev = std::env::var("BLS381_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut p: *mut Bls381Ctx = std::ptr::null_mut();
        check(unsafe { bls381_ctx_create(&mut p, dev) });
        CtxHandle(p)
    })
    .0
}

// ---- arkworks values <-> limb buffers -------------------------------------------------------------
#[inline]
pub fn fq_limbs(x: &Fq) -> [u64; 6] {
    (x.0).0 // Montgomery form, R = 2^384: what the C ABI takes
}
#[inline]
pub fn fq_from_limbs(l: &[u64]) -> Fq {
    Fq::new_unchecked(BigInt::new([l[0], l[1], l[2], l[3], l[4], l[5]]))
}
pub fn push_fq2(buf: &mut Vec<u64>, x: &Fq2) {
    buf.extend_from_slice(&fq_limbs(&x.c0));
    buf.extend_from_slice(&fq_limbs(&x.c1));
}
pub fn fq2_from_limbs(l: &[u64]) -> Fq2 {
    Fq2::new(fq_from_limbs(&l[0..6]), fq_from_limbs(&l[6..12]))
}
pub fn push_g1(buf: &mut Vec<u64>, p: &G1Affine) {
    buf.extend_from_slice(&fq_limbs(&p.x));
    buf.extend_from_slice(&fq_limbs(&p.y));
}
pub fn push_g2(buf: &mut Vec<u64>, q: &G2Affine) {
    push_fq2(buf, &q.x);
    push_fq2(buf, &q.y);
}
pub fn fq12_to_limbs(f: &Fq12) -> [u64; 72] {
    let mut v = Vec::with_capacity(72);
    for c6 in [&f.c0, &f.c1] {
        for c2 in [&c6.c0, &c6.c1, &c6.c2] {
            push_fq2(&mut v, c2);
        }
    }
    v.try_into().unwrap()
}
pub fn fq12_from_limbs(l: &[u64]) -> Fq12 {
    let c = |i: usize| fq2_from_limbs(&l[12 * i..12 * i + 12]);
    Fq12::new(Fq6::new(c(0), c(1), c(2)), Fq6::new(c(3), c(4), c(5)))
}

/// (g1 limbs, g2 limbs, infinity flags) of n pairs; flag bit 0 = P is infinity, bit 1 = Q is infinity.
pub fn flatten_pairs(ps: &[G1Affine], qs: &[(G2Affine, bool)]) -> (Vec<u64>, Vec<u64>, Vec<u8>) {
    assert_eq!(ps.len(), qs.len(), "zip_eq: length mismatch"); // miller_loop.rs:129
    let (mut g1, mut g2, mut inf) = (Vec::with_capacity(12 * ps.len()), Vec::with_capacity(24 * ps.len()), Vec::with_capacity(ps.len()));
    for (p, (q, q_inf)) in ps.iter().zip(qs) {
        push_g1(&mut g1, p);
        push_g2(&mut g2, q);
        inf.push(p.infinity as u8 | ((*q_inf || q.infinity) as u8) << 1);
    }
    (g1, g2, inf)
}
