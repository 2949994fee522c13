/* C ABI of the MI355X (gfx950) batched BLS12-381 pairing engine.
 *
 * The reference (a Rust crate) has no FFI of its own; its boundary is the public Rust API of
 * src/native/.  Each entry point below names the Rust item it replaces.  A Rust caller keeps the
 * reference's signatures and forwards to these functions.
 *
 * Data layouts (all little-endian):
 *   Fq    6 x u64, Montgomery form with R = 2^384: the in-memory limbs of ark_ff's Fq
 *         (BigInt<6>), so no conversion happens on either side.  The limbs must be a canonical
 *         residue (below p), as ark_ff's always are: a pairing, Miller-loop, final-exponentiation or
 *         tower call (everything above the witness section below) that is handed a value >= p
 *         returns BLS381_ERR_INVALID once its work has completed (outputs unspecified); the limbs
 *         entry points report such values through their err bytes; the witness-trace calls reduce
 *         their two input points mod p without a check.
 *   G1    {x, y}                      12 u64  (affine; infinity is carried by inf_flags)
 *   G2    {x.c0, x.c1, y.c0, y.c1}    24 u64
 *   Fq12  72 u64 in ark tower order   c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1
 *   inf_flags  one byte per pair: bit 0 = P is infinity, bit 1 = Q is infinity (may be NULL)
 *   "limbs" entry points use the circuit's witness format instead: canonical (non-Montgomery)
 *   values, 12 x u32 little-endian per Fq (src/fields/fq_target.rs:280-293), Fq12 as 12 Fq in
 *   MyFq12 (w-power) order (src/utils/helpers.rs:14-44).
 *
 * Every function returns 0 on success and a negative code on failure; bls381_last_error() gives a
 * thread-local message.  Nothing aborts or throws across this boundary.  A context is bound to
 * one GPU; calls on one context are serialised internally, so it may be shared by host threads.
 * Buffers are caller-owned host memory unless the name ends in _dev (device memory on the
 * context's GPU, for callers that keep data resident in HBM).
 */
#ifndef BLS381_HIP_H
#define BLS381_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bls381_ctx bls381_ctx;

enum {
    BLS381_OK = 0,
    BLS381_ERR_INVALID = -1,   /* bad argument (NULL pointer, n too large, ...) */
    BLS381_ERR_NO_DEVICE = -2, /* no usable gfx950 device / HIP runtime failure at start-up */
    BLS381_ERR_HIP = -3,       /* a HIP call failed; see bls381_last_error() */
    BLS381_ERR_ZERO = -4,      /* an input that the reference rejects by panicking (zero divisor) */
    BLS381_ERR_POINT = -5,     /* an encoded point was rejected; the per-point status says why */
};

/* Per-point status of the decoding calls. */
enum {
    BLS381_POINT_OK = 0,
    BLS381_POINT_INFINITY = 1,        /* a valid encoding of the point at infinity */
    BLS381_POINT_BAD_ENCODING = 2,    /* flag bits inconsistent, uncompressed form, or x >= p */
    BLS381_POINT_NOT_ON_CURVE = 3,    /* x^3 + b is not a square */
    BLS381_POINT_NOT_IN_SUBGROUP = 4, /* on the curve but outside the order-r subgroup */
};

/* Creates a context on HIP device `device_id`.  Builds and uploads the step programs. */
int bls381_ctx_create(bls381_ctx** out, int device_id);
/* Same, choosing the lane-group width K (6, 8, 9, 12 or 16 lanes per pairing) and the number of
 * LDS slots per group; 0 selects the defaults. */
int bls381_ctx_create_ex(bls381_ctx** out, int device_id, int lanes_per_group, int lds_slots);
void bls381_ctx_destroy(bls381_ctx* ctx);
const char* bls381_last_error(void);

/* multi_miller_loop_native([G1Prepared(p)], [G2Prepared::from(q)]) for each of n independent
 * pairs (src/native/miller_loop.rs:121-162).  A pair with an infinity flag yields Fq12::ONE,
 * as the reference's filter at :130-136 does. */
int bls381_miller_loop_batch(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags,
                             size_t n, uint64_t* out_fq12);
/* final_exponentiation (src/native/final_exponentiation.rs:74-78) of n Fq12 values given in ark
 * order.  is_zero_err (may be NULL) receives 1 for an item equal to zero, on which the reference
 * panics (division at :13); such an item's output is unspecified. Returns BLS381_ERR_ZERO if any. */
int bls381_final_exp_batch(bls381_ctx*, const uint64_t* in_fq12, size_t n, uint64_t* out_fq12,
                           uint8_t* is_zero_err);
/* native_pairing([p], [q]) for each of n independent pairs (src/native/pairing.rs:8-13).  The values are the
 * reference's; the Miller value formed on the way (never returned) differs from bls381_miller_loop_batch's by a factor
 * in Fq, which the final exponentiation removes: the G2 doublings run without their halvings.  The same holds for the
 * two-pair pairing, the product-form pairing (bls381_multi_pairing, bls381_multi_ctx_multi_pairing) and the signature
 * checks; every call that returns Miller values follows the reference's formulas. */
int bls381_pairing_batch(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags, size_t n,
                         uint64_t* out_fq12);
/* multi_miller_loop_native over all n pairs at once: one Fq12 (src/native/miller_loop.rs:121-162;
 * the product over chunks at :155 equals the product of the per-pair values). */
int bls381_multi_miller_loop(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags,
                             size_t n_pairs, uint64_t out_fq12[72]);
/* native_pairing over all n pairs: final_exponentiation of the above.  *is_one = 1 iff the result
 * is Fq12::ONE (the aggregate-signature check). */
int bls381_multi_pairing(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags,
                         size_t n_pairs, uint64_t out_fq12[72], int* is_one);
/* Product of per-rank partial Miller values followed by the final exponentiation: the step after
 * the gather in a multi-GPU multi-pairing.  partials: n_partials x 72 u64. */
int bls381_finish_multi_pairing(bls381_ctx*, const uint64_t* partials, size_t n_partials, uint64_t out_fq12[72],
                                int* is_one);
/* native_pairing(&[pa, pb], &[qa, qb]) for each of n independent items: two pairs share one Miller
 * accumulator, as multi_miller_loop_native runs a chunk of pairs (src/native/miller_loop.rs:139-156),
 * then one final exponentiation (src/native/pairing.rs:8-13).  This is the shape of a signature
 * check, e(pk, H(m)) * e(-G1, sig) == 1: is_one (may be NULL) receives 1 per item whose result is
 * Fq12::ONE; out_fq12 may be NULL when only is_one is wanted.  inf_flags (may be NULL): one byte per
 * item, bit 0 pa / bit 1 qa / bit 2 pb / bit 3 qb is infinity; such a pair is left out of the
 * product, as the filter at :130-136 does.  The first form stops before the final exponentiation. */
int bls381_miller_loop_two_pairs_batch(bls381_ctx*, const uint64_t* g1a, const uint64_t* g2a, const uint64_t* g1b,
                                       const uint64_t* g2b, const uint8_t* inf_flags, size_t n, uint64_t* out_fq12);
int bls381_pairing_two_pairs_batch(bls381_ctx*, const uint64_t* g1a, const uint64_t* g2a, const uint64_t* g1b,
                                   const uint64_t* g2b, const uint8_t* inf_flags, size_t n, uint64_t* out_fq12,
                                   uint8_t* is_one);
/* The same on device memory. */
int bls381_pairing_two_pairs_batch_dev(bls381_ctx*, const uint64_t* g1a_dev, const uint64_t* g2a_dev,
                                       const uint64_t* g1b_dev, const uint64_t* g2b_dev, const uint8_t* inf_flags_dev,
                                       size_t n, uint64_t* out_fq12_dev, uint8_t* is_one_dev);
/* G2Prepared::from(q).ell_coeffs for n points (src/native/miller_loop.rs:90-119):
 * out_coeffs = n x 68 x 3 Fq2 = n x 68 x 36 u64. */
int bls381_g2_prepare_batch(bls381_ctx*, const uint64_t* g2, size_t n, uint64_t* out_coeffs);
/* ell(f, coeffs, p) (src/native/miller_loop.rs:185-194) for n independent triples.
 * f: n x 72 u64 (in and out), coeffs: n x 36 u64, g1: n x 12 u64. */
int bls381_ell_batch(bls381_ctx*, const uint64_t* f_in, const uint64_t* coeffs, const uint64_t* g1, size_t n,
                     uint64_t* f_out);
/* easy_part_native (src/native/final_exponentiation.rs:8-19) and hard_part_native (:44-72, any
 * Fq12, generic squarings) as separate batch calls; ark order in and out. */
int bls381_easy_part_batch(bls381_ctx*, const uint64_t* in_fq12, size_t n, uint64_t* out_fq12, uint8_t* is_zero_err);
int bls381_hard_part_batch(bls381_ctx*, const uint64_t* in_fq12, size_t n, uint64_t* out_fq12);
/* frobenius_map_native(a, power) for any power, reduced mod 12 as the reference does
 * (src/native/fin_exp_helpers.rs:10-47); power 0 is the identity. */
int bls381_frobenius_batch(bls381_ctx*, const uint64_t* in_fq12, unsigned power, size_t n, uint64_t* out_fq12);
/* conjugate_fp12 (src/native/fin_exp_helpers.rs:132-142): the coefficients of the odd powers of w change sign. */
int bls381_conjugate_batch(bls381_ctx*, const uint64_t* in_fq12, size_t n, uint64_t* out_fq12);
/* pow_native(a, exp) (src/native/fin_exp_helpers.rs:60-82) for any Fq12 a (generic squarings) and one exponent for
 * the whole batch: exp = exp_words (1..6) little-endian 64-bit words, the reference's Vec<u64>.  As in the
 * reference an exponent of zero returns a itself (its loop never starts).  bls381_pow_x_batch is
 * pow_native(a, vec![BLS_X]), the call the hard part makes five times (final_exponentiation.rs:46-63). */
int bls381_pow_batch(bls381_ctx*, const uint64_t* in_fq12, const uint64_t* exp, size_t exp_words, size_t n,
                     uint64_t* out_fq12);
int bls381_pow_x_batch(bls381_ctx*, const uint64_t* in_fq12, size_t n, uint64_t* out_fq12);
/* Fq12 product (ark Fq12 `*`, equal to MyFq12::mul, src/utils/helpers.rs:90-152). */
int bls381_fq12_mul_batch(bls381_ctx*, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out_fq12);

/* Fq12InverseGenerator::run_once (src/fields/fq12_target.rs:374-397) for n values:
 * x and inv are n x 144 u32 canonical limbs in MyFq12 order.  err (may be NULL): per item,
 * 1 = a limb-decoded coefficient >= p, 2 = x is zero (both panic in the reference). */
int bls381_fq12_inverse_limbs_batch(bls381_ctx*, const uint32_t* x, size_t n, uint32_t* inv, uint8_t* err);

/* Witness trace of the single-pair pairing circuit (src/pairing.rs:13-20 after
 * G2PreparedTarget::from, src/curves/g2.rs:148-178): the value of every non-native Fq operation
 * in builder call order, plus the 12 values written by Fq12InverseGenerator at the point where
 * Fq12Target::inv is called.  Rows are 12 x u32 canonical limbs.
 *   bls381_pairing_witness_rows()  number of rows per proof (141,376)
 *   bls381_pairing_witness_kinds() fills kinds[rows]: 0 add, 1 sub, 2 mul, 3 neg, 4 inv, 5 generator output
 *   trace: n_proofs x rows x 12 u32 */
size_t bls381_pairing_witness_rows(void);
int bls381_pairing_witness_kinds(uint8_t* kinds, size_t capacity);
int bls381_pairing_witness_batch(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, size_t n_proofs, uint32_t* trace);
/* The same with g1, g2 and trace in device memory: the trace is written in place (6.79 MB per proof;
 * the copy to the host is what dominates the host-pointer form). */
int bls381_pairing_witness_batch_dev(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, size_t n_proofs,
                                     uint32_t* trace);

/* The trace together with the auxiliary witness values of the non-native gadgets behind FqTarget's add / sub / mul /
 * neg / inv (src/fields/fq_target.rs:144-192; the gadgets and their generators live in plonky2_ecdsa at the
 * InternetMaximalism fork's rev 541e127, Cargo.toml:8-10, which is not part of the reference tree: the VALUES below
 * are determined by each row's operands, their limb LAYOUT in the fork is an assumption - parity unpinned).
 * Any of the three outputs may be NULL.
 *   mul_quot  n_proofs x bls381_pairing_witness_mul_rows() x 24 u32: per mul row, in row order, the quotient
 *             (a * b - r) / p of the product by the modulus (what NonNativeMultiplicationGenerator's div_rem yields
 *             next to the remainder r that is the row's value); below p, so limbs 12..23 are zero
 *   carry     n_proofs x rows bytes: add rows 1 iff a + b >= p (the modulus was subtracted), sub rows 1 iff a < b (the
 *             modulus was added), neg rows 1 iff a != 0 (neg is 0 - a), other rows 0
 *   inv_quot  n_proofs x 24 u32: for the one inv row (inv_nonnative(2), src/curves/g2.rs:149-151) (a * a^-1 - 1) / p */
size_t bls381_pairing_witness_mul_rows(void);
int bls381_pairing_witness_aux_batch(bls381_ctx*, const uint64_t* g1, const uint64_t* g2, size_t n_proofs, uint32_t* trace,
                                     uint32_t* mul_quot, uint8_t* carry, uint32_t* inv_quot);

/* Device-resident variants: all pointers are device memory on the context's GPU; they run on the
 * context's stream and return after the work has completed. */
int bls381_pairing_batch_dev(bls381_ctx*, const uint64_t* g1_dev, const uint64_t* g2_dev, const uint8_t* inf_flags_dev,
                             size_t n, uint64_t* out_fq12_dev);
int bls381_miller_loop_batch_dev(bls381_ctx*, const uint64_t* g1_dev, const uint64_t* g2_dev,
                                 const uint8_t* inf_flags_dev, size_t n, uint64_t* out_fq12_dev);
int bls381_final_exp_batch_dev(bls381_ctx*, const uint64_t* in_fq12_dev, size_t n, uint64_t* out_fq12_dev,
                               uint8_t* is_zero_err_dev);

/* ---- several GPUs behind one handle ------------------------------------------------------------
 * The reference is single-threaded host code with no device notion (/root/reference/src/native/miller_loop.rs:139 runs
 * its chunks sequentially); a drop-in for a multi-GPU node keeps one call per batch and lets the library spread it.
 * bls381_multi_ctx_create builds one context per entry of device_ids (an id may be listed more than once: its
 * contexts then share that GPU).  Every call runs one host thread per device for its duration and gives device k the
 * contiguous shard [k n / G, (k + 1) n / G) of the batch; results land in the caller's buffers at the items' own
 * positions, so the output is identical to the single-context call's.  Independent items need no exchange between
 * devices.  The product forms (multi_miller_loop_native / native_pairing over all pairs,
 * src/native/miller_loop.rs:121-162, src/native/pairing.rs:8-13) reduce each shard to one Fq12 on its device, gather
 * the G partials (576 bytes each) on the host and multiply them - and run the one final exponentiation - on the
 * first device.  Host pointers throughout. */
typedef struct bls381_multi_ctx bls381_multi_ctx;
int bls381_multi_ctx_create(bls381_multi_ctx** out, const int* device_ids, int n_devices);
void bls381_multi_ctx_destroy(bls381_multi_ctx*);
int bls381_multi_ctx_devices(const bls381_multi_ctx*);
int bls381_multi_ctx_pairing_batch(bls381_multi_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags, size_t n,
                                   uint64_t* out_fq12);
int bls381_multi_ctx_miller_loop_batch(bls381_multi_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags,
                                       size_t n, uint64_t* out_fq12);
int bls381_multi_ctx_final_exp_batch(bls381_multi_ctx*, const uint64_t* in_fq12, size_t n, uint64_t* out_fq12,
                                     uint8_t* is_zero_err);
int bls381_multi_ctx_multi_miller_loop(bls381_multi_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags,
                                       size_t n_pairs, uint64_t out_fq12[72]);
int bls381_multi_ctx_multi_pairing(bls381_multi_ctx*, const uint64_t* g1, const uint64_t* g2, const uint8_t* inf_flags,
                                   size_t n_pairs, uint64_t out_fq12[72], int* is_one);
int bls381_multi_ctx_pairing_witness_batch(bls381_multi_ctx*, const uint64_t* g1, const uint64_t* g2, size_t n_proofs,
                                           uint32_t* trace);
int bls381_multi_ctx_verify_batch(bls381_multi_ctx*, const uint8_t* pk48, const uint8_t* msg96, const uint8_t* sig96, size_t n,
                                  int check_subgroup_mask, uint8_t* verdict, uint8_t* status);

/* The context's workspace grows on demand (a call larger than any before it frees and reallocates device buffers
 * inside the call).  bls381_ctx_reserve does that growing ahead of time for pairing, Miller-loop and
 * final-exponentiation calls of up to max_items items (one pair or one Fq12 each; batches are cut into chunks of
 * 65,536, so larger values reserve the same), and builds the program variants small batches use: the first large call
 * then costs what every later one does.  No counterpart in the reference (host memory there is the allocator's). */
int bls381_ctx_reserve(bls381_ctx*, size_t max_items);

/* bls381_pairing_batch_dev for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, code that must not disturb an allocator's state).  The call runs on the context's own stream and returns
 * after synchronising it, so it is not itself capturable into a HIP graph: it serves the code around a capture that may
 * not allocate.  No counterpart in the reference.  Same kernels and results as
 * bls381_pairing_batch_dev, but the workspace (slot arrays, carry and spill arrays) is a fixed 32 MiB array that
 * the library holds in each GPU's memory from load to process exit; the call makes no hipMalloc / hipFree.
 * n is at most BLS381_NOALLOC_MAX_PAIRS (larger n: BLS381_ERR_INVALID, nothing runs).  All calls on one GPU share
 * that array, from every context on it: two such calls that overlap in time on one GPU race, the caller
 * serialises them (calls on one context already are).  Nothing in the array needs clearing between calls:
 * every location is written before it is read. */
enum { BLS381_NOALLOC_MAX_PAIRS = 4096 };
int bls381_pairing_batch_dev_noalloc(bls381_ctx*, const uint64_t* g1_dev, const uint64_t* g2_dev,
                                     const uint8_t* inf_flags_dev, size_t n, uint64_t* out_fq12_dev);

/* Introspection used by the tests and the benchmark. */
/* Builds and schedules the step program `which` ("miller", "millerfq", "miller2", "fe1", "fe2", "fe2s0" .. "fe2s5", "g2prep", "g1sub", "g2sub", "ell", "fq12mul",
 * "easy2", "hardgen", "inv1", "inv2", "conj", "frob0" .. "frob11", "pow:<16 hex digits per exponent word>") for groups of K lanes and copies it into
 * caller buffers: hdrs = nsteps x 2 u32, descs = nsteps x K x 12 u32.  info[8] receives
 * {K, nsteps, lds_slots, tmp_slots, n_in, n_out, multiply-accumulates executed (300 per product, 444 per two-product sum), n_lin}.  Pass NULL buffers to query sizes.
 * Host-only: needs no GPU. */
int bls381_debug_program(const char* which, int lanes_per_group, int lds_slots, uint32_t* hdrs, uint32_t* descs,
                         size_t capacity_steps, uint32_t info[8]);
/* Element-wise Montgomery products a[i] * b[i] / 2^384 mod p of n Fq pairs (12 x u32 limbs each)
 * with the hand-scheduled multiplier (out) and the plain-C form it is cross-checked against
 * (out_plain). */
int bls381_debug_fp_mul(bls381_ctx*, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out,
                        uint32_t* out_plain);
/* Element-wise Fq inversion of n <= 65,536 values in boundary form (12 x u32, Montgomery 2^384):
 * out = a^-1 in the same form, 0 for a = 0 with is_zero[i] = 1.  The inversion kernel that
 * Fq12::inverse (/root/reference/src/native/final_exponentiation.rs:9, through ark-ff) bottoms out
 * in, exposed for direct testing. */
int bls381_debug_fp_inverse(bls381_ctx*, const uint32_t* a, size_t n, uint32_t* out, uint8_t* is_zero);
/* The squaring chains of the hard part (/root/reference/src/native/final_exponentiation.rs:44-72 calls pow_native five
 * times, /root/reference/src/native/fin_exp_helpers.rs:60-82: 63 squarings each) run as a kernel of their own that keeps
 * a^(2^k) for the set bits k of |x| = 0xd201000000010000 (16, 48, 57, 60, 62, 63): this runs it alone on n <= 65,536
 * values of the cyclotomic subgroup.  in: n x 12 Fq (ark tower order) as 12 x u32 canonical words carrying the
 * Montgomery factor 2^392 (the kernels' internal factor; not the 2^384 of the entry points above), out: n x 6 x 12 Fq in
 * the same form, the kept powers in ascending order of k.  bls381_last_timing()[2] then holds the kernel's time. */
int bls381_debug_cyc_chain(bls381_ctx*, const uint32_t* in, size_t n, uint32_t* out);
/* Pairing and final-exponentiation calls of more than 8,192 items run the second part of the final exponentiation as
 * six step programs with that kernel between them (BLS381_FE2_CHAIN=0 in the environment keeps the single program,
 * =1 takes the chained form for every size; same values either way).  After such a call: ms[0] = its five chain
 * launches together, ms[1] = its six programs together (last 65,536-item chunk), *used = 1; *used = 0 if the single
 * program ran. */
int bls381_last_chain_timing(bls381_ctx*, float ms[2], int* used);
/* The witness kernel keeps `cache_rows` (0..128) trace rows per proof in LDS, filled and read according to a plan
 * made on the host for the number of lanes that serve one proof (a divisor of 64).  Builds that plan and replays it:
 * out = {operand reads of trace rows, reads the plan serves from LDS, plan violations (must be 0), operations}.
 * Host-only: needs no GPU. */
int bls381_debug_witness_cache_plan(int lanes_per_proof, int cache_rows, uint64_t out[4]);
/* The operations the witness kernel runs in one phase (0: before the Fq12 inversion, 1: after), in its order:
 * ops[3 i .. 3 i + 2] = {dst, a, b} of operation i (dst: row | kind << 28; a, b: space << 30 | index; spaces 0 row,
 * 1 constant, 2 circuit input), levels = offsets of the dependency levels into them.  Rows from the trace's row count
 * on are the kernel's own (products' left operands times 2^392); constants 24..47 are constants 0..23 times 2^392,
 * constant 48 is 2^784; inputs 6..11 are inputs 0..5 times 2^392; a product is a * b / 2^392.
 * sizes = {operations, levels + 1, trace rows, kernel's own rows}; with null buffers only sizes is filled.
 * No counterpart in the reference: test access to the program the trace is computed by. */
int bls381_debug_witness_ops(int phase, uint32_t* ops, size_t ops_capacity, uint32_t* levels, size_t levels_capacity,
                             uint64_t sizes[4]);
/* The interpreter's constant table: C_COUNT rows of 16 u32 (14 digits of 28 bits, zero padded). */
int bls381_debug_constants(uint32_t* out, size_t capacity_fp);
/* ---- the other witness generators of the reference --------------------------------------------
 * Canonical (non-Montgomery) 12 x u32 limbs in and out, as the generators read them with
 * get_biguint_target and write them with set_biguint_target.  err (optional) gets one byte per
 * element; a non-zero byte marks an element on which the reference panics (or limbs >= p), its
 * output is zero and the call returns BLS381_ERR_ZERO after filling all outputs.
 *   FqSqrtGenerator   /root/reference/src/fields/fq_target.rs:296-323   x, sgn -> the root whose parity bit is sgn
 *   Fq2SqrtGenerator  /root/reference/src/fields/fq2_target.rs:401-428  x (c0, c1), sgn -> the root with sgn0_fq2 = sgn
 *   Fq2InverseGenerator  fq2_target.rs:347-369   x -> 1/x, zero -> zero (no error)
 *   Fq6InverseGenerator  /root/reference/src/fields/fq6_target.rs:350-371   x in MyFq6 order
 *                        [c0.c0, c1.c0, c2.c0, c0.c1, c1.c1, c2.c1] -> 1/x in the same order; zero is an error */
int bls381_fq_sqrt_limbs_batch(bls381_ctx*, const uint32_t* x, const uint8_t* sgn, size_t n, uint32_t* out, uint8_t* err);
int bls381_fq2_sqrt_limbs_batch(bls381_ctx*, const uint32_t* x, const uint8_t* sgn, size_t n, uint32_t* out, uint8_t* err);
int bls381_fq2_inverse_limbs_batch(bls381_ctx*, const uint32_t* x, size_t n, uint32_t* out, uint8_t* err);
int bls381_fq6_inverse_limbs_batch(bls381_ctx*, const uint32_t* x, size_t n, uint32_t* out, uint8_t* err);

/* ---- compressed points and aggregation: the step before the path in BLS signature verification --
 * The reference takes ark-ec points (/root/reference/src/native/pairing.rs:8-13); its callers decode
 * them with ark-bls12-381 0.4.0 (/root/reference/Cargo.toml:13) G1Affine / G2Affine
 * deserialize_compressed and encode them with serialize_compressed.  These calls do that for whole
 * batches: 48 bytes per G1 point (big-endian x), 96 per G2 point (big-endian x.c1 then x.c0), top
 * bits of byte 0 = 0x80 compressed, 0x40 infinity, 0x20 y is the lexicographically larger root.
 * Decoded points come out in the G1 / G2 layout above, ready for the pairing calls; status (may be
 * NULL) gets one BLS381_POINT_* byte per point, rejected points are zero and the call returns
 * BLS381_ERR_POINT after filling everything.  A point at infinity is status 1 with a zero point:
 * pass it on through inf_flags.  check_subgroup != 0 also runs the order-r membership test that
 * deserialize_compressed performs (Validate::Yes); 0 corresponds to deserialize_compressed_unchecked
 * plus the on-curve test.  (The membership tests run as step programs over lane groups after the decoder: 65,536
 * points decoded and tested in 2.9 ms (G1) / 4.2 ms (G2).) */
int bls381_g1_decompress_batch(bls381_ctx*, const uint8_t* in48, size_t n, int check_subgroup, uint64_t* g1,
                               uint8_t* status);
int bls381_g2_decompress_batch(bls381_ctx*, const uint8_t* in96, size_t n, int check_subgroup, uint64_t* g2,
                               uint8_t* status);
/* inf (may be NULL): one byte per point, non-zero = the point at infinity. */
int bls381_g1_compress_batch(bls381_ctx*, const uint64_t* g1, const uint8_t* inf, size_t n, uint8_t* out48);
int bls381_g2_compress_batch(bls381_ctx*, const uint64_t* g2, const uint8_t* inf, size_t n, uint8_t* out96);
/* Sums of n_groups groups of group_size points each (ark-ec `G1Projective::sum` / the `+` of public
 * keys or signatures before /root/reference/src/native/pairing.rs:8-13 sees them): points laid out
 * group after group, inf as above; out gets one affine point per group, out_inf its infinity flag. */
int bls381_g1_aggregate_batch(bls381_ctx*, const uint64_t* g1, const uint8_t* inf, size_t n_groups, size_t group_size,
                              uint64_t* out, uint8_t* out_inf);
int bls381_g2_aggregate_batch(bls381_ctx*, const uint64_t* g2, const uint8_t* inf, size_t n_groups, size_t group_size,
                              uint64_t* out, uint8_t* out_inf);

/* n signature checks from bytes (minimal-public-key-size BLS: keys in G1, signatures and message points in
 * G2), the whole chain on the device: decode the three encodings as above, reject a key at infinity, then
 * native_pairing(&[pk, -G1::generator()], &[msg, sig]) == Fq12::ONE (/root/reference/src/native/pairing.rs:8-13)
 * through the two-pair call.  msg96 is the compressed hash-to-curve output of the message (hashing itself is not
 * done here).  A message point at infinity is rejected like a key at infinity (verdict 0): no message hashes to it, and
 * together with a signature at infinity it would leave an empty product.  verdict: one byte per item, 1 = the signature verifies.  status (may be NULL): 3 bytes per item,
 * the BLS381_POINT_* status of key, message point and signature.  Malformed input is a verdict of 0, not an
 * error return.  check_subgroup_mask says which of the three inputs get the order-r membership test: keys and
 * signatures come from outside; message points a caller has hashed itself are in the subgroup by construction
 * and can skip it (BLS381_CHECK_KEYS | BLS381_CHECK_SIGNATURES). */
enum { BLS381_CHECK_KEYS = 1, BLS381_CHECK_MESSAGES = 2, BLS381_CHECK_SIGNATURES = 4 };
int bls381_verify_batch(bls381_ctx*, const uint8_t* pk48, const uint8_t* msg96, const uint8_t* sig96, size_t n,
                        int check_subgroup_mask, uint8_t* verdict, uint8_t* status);

/* One signature by n_keys signers over one message (FastAggregateVerify; the 512-key sync-committee check): keys
 * decoded and summed on the device, then native_pairing(&[sum, -G1::generator()], &[msg, sig]) == Fq12::ONE.
 * *valid = 1 iff every key decodes to a finite point, the SUM of the keys is not the point at infinity (KeyValidate
 * on the aggregate, as FastAggregateVerify's CoreVerify step does: keys such as {pk, -pk} must verify nothing),
 * message point (finite) and signature decode, and the product is one; key_status (may be NULL) receives the n_keys decoder statuses.  check_subgroup_mask as above. */
int bls381_fast_aggregate_verify(bls381_ctx*, const uint8_t* pk48, size_t n_keys, const uint8_t* msg96,
                                 const uint8_t* sig96, int check_subgroup_mask, int* valid, uint8_t* key_status);
/* n_checks such checks at once: pk48 holds keys_per_check keys per check, check after check; one message point
 * and one signature per check; verdict one byte per check; key_status (may be NULL) one byte per key. */
int bls381_fast_aggregate_verify_batch(bls381_ctx*, const uint8_t* pk48, size_t n_checks, size_t keys_per_check,
                                       const uint8_t* msg96, const uint8_t* sig96, int check_subgroup_mask,
                                       uint8_t* verdict, uint8_t* key_status);

/* ---- one G2 point against many G1 points ------------------------------------------------------
 * G2Prepared (/root/reference/src/native/miller_loop.rs:13-16, 90-119) holds the 68 line coefficients
 * of Q so that they are computed once and reused by every Miller loop against that Q (:139-156).
 * bls381_g2_prepared_create computes them on the device (same program as bls381_g2_prepare_batch)
 * and keeps them there in the kernels' internal number format; g2 = {x.c0, x.c1, y.c0, y.c1} as
 * 24 u64, ignored when q_is_infinity != 0 (G2Prepared::infinity: every pair is then skipped and
 * yields one).  The handle belongs to the context's device; destroy it before the context. */
typedef struct bls381_g2_prepared bls381_g2_prepared;
int bls381_g2_prepared_create(bls381_ctx*, const uint64_t* g2, int q_is_infinity, bls381_g2_prepared** out);
void bls381_g2_prepared_destroy(bls381_g2_prepared*);
/* multi_miller_loop_native([p_i], [q]) for each i (conjugated, before the final exponentiation;
 * miller_loop.rs:121-162 with b = the one prepared point), and native_pairing([p_i], [q])
 * (pairing.rs:8-13).  p_is_infinity: optional byte per element, non-zero = P_i is the point at
 * infinity (the pair is skipped: result one).  The _dev form takes device pointers. */
int bls381_miller_loop_fixed_q_batch(bls381_ctx*, const bls381_g2_prepared* q, const uint64_t* g1,
                                     const uint8_t* p_is_infinity, size_t n, uint64_t* out_fq12);
int bls381_pairing_fixed_q_batch(bls381_ctx*, const bls381_g2_prepared* q, const uint64_t* g1,
                                 const uint8_t* p_is_infinity, size_t n, uint64_t* out_fq12);
int bls381_pairing_fixed_q_batch_dev(bls381_ctx*, const bls381_g2_prepared* q, const uint64_t* g1,
                                     const uint8_t* p_is_infinity, size_t n, uint64_t* out_fq12);

/* Per-step-kind shader-clock accounting of program `which` ("miller", "fe2", ...) run over n
 * groups of zero inputs: out[k] cycles and out[16 + k] step counts of the grid's first wave, k = 0
 * product, 1 sum, 2 sum + carry sweep, 3 two-product, 4.. copies by kind (in, fill, constant, spill,
 * out), 9 sum + reduction; out[15] = nanoseconds of the same launch repeated without the accounting.
 * Names starting with "t_" run synthetic single-kind programs ("t_lin4", "t_dot2", "t_spill", ...).
 * Tuning aid; no counterpart in the reference. */
int bls381_debug_profile(bls381_ctx*, const char* which, size_t n, uint64_t out[32]);
/* Timing of the most recent pairing / Miller-loop / final-exponentiation batch call on this context
 * (its last 65,536-element chunk), from HIP events on the context's stream, in milliseconds:
 * ms[0] = Miller-loop kernel, ms[1] = final exponentiation part 1 + Fp inversion kernel,
 * ms[2] = final exponentiation part 2 kernel, ms[3] = whole device-side call incl. layout conversion. */
int bls381_last_timing(bls381_ctx*, float ms[4]);
/* Sustained v_mad_u64_u32 rate of the device in multiply-accumulates per second (calibration kernel). */
int bls381_calibrate_mac_rate(bls381_ctx*, double* mac_per_s);

#ifdef __cplusplus
}
#endif
#endif /* BLS381_HIP_H */
