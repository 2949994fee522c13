// Witness trace of the single-pair pairing circuit: the value of every non-native Fq operation the
// reference's circuit builders create, in builder call order.
//
// Host side: a symbolic replay of the builder call sequence
//   G2PreparedTarget::from   /root/reference/src/curves/g2.rs:148-178 (steps :59-129)
//   multi_miller_loop        /root/reference/src/miller_loop.rs:19-79
//   final_exponentiation     /root/reference/src/final_exponentiation.rs:12-64
//   frobenius_map/pow_target /root/reference/src/final_exponentiation_helpers.rs:14-85
//   Fq12/Fq6/Fq2 gadgets     /root/reference/src/fields/{fq12,fq6,fq2}_target.rs
// producing one record per row (operation kind + operand references), level-scheduled so that a
// workgroup can evaluate each dependency level in parallel.
// Device side: one workgroup per proof walks the levels; rows are canonical 12 x u32 limbs
// (/root/reference/src/fields/fq_target.rs:280-293).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace bls381 {

enum WitnessKind : uint8_t { WK_ADD = 0, WK_SUB = 1, WK_MUL = 2, WK_NEG = 3, WK_INV = 4, WK_GEN = 5,
                             WK_NOP = 7 };  // (only in the padded operation list the kernel walks: a lane without a row in a round)

// operand reference: bits 30..31 = space (0 trace row, 1 constant, 2 circuit input), bits 0..17 = index.
// Row indices from n_rows on name rows the kernel keeps beside the trace, in a buffer of its own (the products' left
// operands, see WitnessProgram::n_shadow).
// In the operations handed to the kernel (plan_row_cache) a row reference may also say that the row sits in the
// proof's LDS row cache: bit 25 set, entry number in bits 18..24.  WitnessOp::dst carries the same fields for the
// row an operation produces: bit 25 = the result is kept in the cache entry of bits 18..24 as well.
enum { WREF_ROW = 0u << 30, WREF_CONST = 1u << 30, WREF_INPUT = 2u << 30, WREF_MASK = (1u << 30) - 1 };
enum : uint32_t { WROW_MASK = (1u << 18) - 1, WCACHE_SLOT_SHIFT = 18, WCACHE_SLOT_MASK = 0x7f, WCACHE_FLAG = 1u << 25, WCACHE_MAX_ROWS = 128,
                  WDRAIN_FLAG = 1u << 26,    // in WitnessOp::dst: finish every earlier store before this operation's reads (see WitnessProgram::rounds)
                  WBARRIER_FLAG = 1u << 27,  // in WitnessOp::dst, on every operation of a round: a dependency level ends with this round
                  WNOSTORE_FLAG = 1u << 31,  // in WitnessOp::dst: a shadow row that every reader finds in the LDS row cache stays out of global memory
                  WKIND_MASK = 7 };          // the kind is (dst >> 28) & WKIND_MASK

struct WitnessOp {
    uint32_t dst;   // row index (bits 0..17) | cache fields | kind << 28
    uint32_t a, b;  // operand references (b unused for neg / inv)
};

// canonical constants the circuit uses (FqTarget::constant values)
enum WitnessConst {
    WC_ZERO = 0, WC_ONE = 1, WC_FOUR = 2, WC_TWO_INV = 3,
    WC_FROB1_BASE = 4,   // + 2*(i-1) + {0,1}, i = 1..5
    WC_FROB2_BASE = 14,  // + 2*(i-1) + {0,1}, i = 1..5 (c1 = 0)
    WC_COUNT = 24,
    WC_TIMES_R = 24,     // + k: constant k times 2^392 (a product's left operand, see WitnessProgram::n_shadow)
    WC_R2 = 48,          // 2^784: the left operand that turns a canonical row into row * 2^392
    WC_TOTAL = 49,
};
// Operands the kernel takes from its instruction stream instead of memory: the constant zero (a copy is an addition of
// zero) and 2^784 (the left operand that makes a shadow row).  The cache plan and its replay count the waits of a wave
// for its loads, so they have to agree with the kernel on what is a load.
inline bool wref_is_immediate(uint32_t ref) { return ref == (WREF_CONST | (uint32_t)WC_ZERO) || ref == (WREF_CONST | (uint32_t)WC_R2); }
// The kernel reads its operations from LDS: a wave keeps two blocks of WOPS_PER_BLOCK operations (128 / lpp rounds
// each) there and fetches the block after next from global memory once per block, see witness_kernel.
enum { WOPS_PER_BLOCK = 128, WOPS_RING_BYTES = 2 * WOPS_PER_BLOCK * 12 };
// circuit inputs per proof: 0..5 canonical {px, py, qx.c0, qx.c1, qy.c0, qy.c1}, 6..11 the same times 2^392
enum { WIN_COUNT = 6, WIN_TOTAL = 12 };

struct WitnessProgram {
    int group = 0;                        // rows of a level that run together, as the level assignment assumed (0: none)
    // How the operations are arranged.  false: in dependency levels (no operation reads a row of its own level; the
    // kernel ends a level with a barrier that also finishes its stores).  true: in rounds of at most `group` operations,
    // each a "level" of its own in level_start, chosen in row order from what is ready, products and other operations
    // in rounds of their own; a round may read what the round before it made, through the proof's LDS row cache, and
    // no barrier separates rounds.  A row read from global memory must have been stored before an earlier wait of the
    // wave for its memory operations: plan_row_cache marks the reads that would come too early with WDRAIN_FLAG.
    bool rounds = false;
    size_t n_rows = 0;
    std::vector<uint8_t> kinds;           // per row
    size_t gen_start = 0;                 // first of the 12 generator-output rows
    uint32_t inv_input_refs[12];          // rows holding x (MyFq12 order) that Fq12InverseGenerator inverts
    // What the kernel runs.  A product row is one Montgomery multiplication on 28-bit digits, left * right / 2^392, so
    // its left operand has to be (a value) * 2^392: a constant or circuit input kept in that form as well, or a "shadow"
    // row = 2^784 * (a trace row) / 2^392, made once by a product operation of its own and used by every product that
    // has the row as a factor (an Fq12Target::mul makes 12 of them for its 144 + 10 products).  Shadow rows have the
    // indices n_rows .. n_rows + n_shadow[0] + n_shadow[1] (phase 0's first) and live in a buffer beside the trace.
    // A product by the constant one is a copy here (an addition of zero), and so is a product of the two factors of an
    // earlier product (x_j * x_i after x_i * x_j in a squaring).
    std::vector<WitnessOp> ops[2];        // phase 0: rows before the generator output, phase 1: rows after
    size_t n_shadow[2] = {0, 0};
    size_t n_ext = 0;                     // n_rows + the shadow rows
    // The circuit's operations themselves, one per row in row order (the auxiliary-value kernel reads these).
    std::vector<WitnessOp> circuit_ops[2];
    std::vector<uint32_t> level_start[2]; // per phase, nlevels + 1 offsets into ops[phase]
    size_t counts[6] = {0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> mul_index;      // per row: its ordinal among the mul rows (0xffffffff for other kinds)
    std::string error;                    // non-empty: the program could not be built (WitnessDevice::run reports it)
};
// optional outputs beside the trace (any may be null): see bls381_pairing_witness_aux_batch
struct WitnessAux {
    uint32_t* mul_quot;  // n x (number of mul rows) x 24
    uint8_t* carry;      // n x rows
    uint32_t* inv_quot;  // n x 24
};
const WitnessProgram& witness_program();
// The operations of one phase with their row-cache fields filled in for `lpp` lanes per proof and `cache_rows` cache
// entries per proof (0: every read from the trace); optionally the number of row reads and how many the cache serves.
std::vector<WitnessOp> plan_row_cache(const WitnessProgram& P, int phase, int lpp, int cache_rows, size_t* reads_out = nullptr,
                                      size_t* hits_out = nullptr);
// The planned operations as the kernel walks them: rounds of exactly `lpp` entries (WK_NOP where a round has fewer
// operations), WBARRIER_FLAG on the last round of every dependency level when the program is arranged in levels.
std::vector<WitnessOp> pad_rounds(const WitnessProgram& P, int phase, const std::vector<WitnessOp>& planned, int lpp);

struct bls381_ctx_fwd;

class WitnessDevice {
   public:
    // Runs n proofs.  g1/g2: host arrays in the ABI layout (Montgomery).  trace: host, n * n_rows * 12 u32.
    // `run_inverse` performs the Fq12 inversion of n values given as device canonical limbs (n x 144 u32, MyFq12
    // order) into device canonical limbs; it is supplied by the context (lane-group programs inv1/inv2).
    typedef int (*inverse_fn)(void* ctx, const uint32_t* x_dev, size_t n, uint32_t* inv_dev);
    // on_device: g1, g2 and trace are device pointers; the trace is then written in place, no staging copy
    int run(hipStream_t stream, void* ctx, inverse_fn inv, const uint64_t* g1, const uint64_t* g2, size_t n,
            uint32_t* trace, std::string& err, bool on_device = false, const WitnessAux* aux = nullptr);
    void release();

   private:
    bool uploaded_ = false;
    int prog_group_ = -1;
    WitnessOp* d_ops_[2] = {nullptr, nullptr};   // as built (the auxiliary-value kernel reads these)
    WitnessOp* d_plan_[2] = {nullptr, nullptr};  // with the row-cache plan for (plan_lpp_, plan_rows_)
    int plan_lpp_ = -1, plan_rows_ = -1;
    size_t n_rounds_[2] = {0, 0};
    void* d_trash_ = nullptr;  size_t trash_bytes_ = 0;
    // large results on their way to the caller's (pageable) memory: two pinned buffers, see copy_out_staged
    void* h_stage_[2] = {nullptr, nullptr};
    hipEvent_t stage_ev_[2] = {nullptr, nullptr};
    int copy_out_staged(hipStream_t stream, void* dst_host, const void* src_dev, size_t bytes, std::string& err);
    uint32_t* d_consts_ = nullptr;
    uint32_t* d_inv_refs_ = nullptr;
    void* d_trace_ = nullptr;  size_t trace_bytes_ = 0;
    void* d_inputs_ = nullptr; size_t inputs_bytes_ = 0;
    void* d_x_ = nullptr;      size_t x_bytes_ = 0;
    void* d_g_ = nullptr;      size_t g_bytes_ = 0;
    void* d_aux_ = nullptr;    size_t aux_bytes_ = 0;
    void* d_shadow_ = nullptr; size_t shadow_bytes_ = 0;
    uint32_t* d_mul_index_ = nullptr;
};

}  // namespace bls381
