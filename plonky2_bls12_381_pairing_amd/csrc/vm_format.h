// Encoding of a lane-group program ("tape") shared by the host-side scheduler
// and the gfx950 interpreter kernel.
//
// A program is executed by groups of K adjacent lanes; one group owns one
// pairing (or one Fq12 value).  Every value is one Fp element (12 x u32,
// Montgomery form) held in an LDS slot private to the group.  A program is a
// list of steps; in one step every lane of the group executes at most one
// operation of the step's type.  All lanes read their operands before any lane
// writes its result, so a destination may alias any operand of the same step.
#pragma once
#include <stdint.h>

namespace bls381 {

enum StepType : uint32_t {
    ST_FMUL = 0,  // dst = a * b           (Montgomery product, canonical result)
    ST_LIN  = 1,  // dst = sum_i (+/-) t_i (canonical result)
    ST_COPY = 2,  // LDS <-> global slot arrays, constant table -> LDS
    ST_DOT2 = 3,  // dst = a * b + c * d   (one Montgomery reduction, canonical result)
};

enum CopyKind : uint32_t {
    CK_IN    = 0,  // LDS[dst] = gin [src]
    CK_FILL  = 1,  // LDS[dst] = gtmp[src]
    CK_CONST = 2,  // LDS[dst] = consts[src]
    CK_SPILL = 3,  // gtmp[dst] = LDS[src]
    CK_OUT   = 4,  // gout[dst] = LDS[src]
};

struct StepHdr {
    uint32_t type;       // StepType
    uint32_t max_terms;  // ST_LIN: bits 0..7 largest term count among the step's ops,
                         //         bits 8..15 rounds of conditional subtraction (2^(r-1) p .. p)
};

// One operation descriptor per (step, lane-in-group).  32 bytes.
//   hdr bit 0      : active
//   hdr bit 1      : ST_DOT2: the op has a second product (else dst = a * b)
//   hdr bit 2      : ST_DOT2: the second product is subtracted; bits 16..23 = k with d <= k * p
//   Idle lanes of multiplier and sum steps carry dst = TRASH slot and operands = ZERO slot (the
//   program's last two LDS slots), so those steps run without per-lane branches.
//   hdr bits 8..15 : ST_LIN term count / ST_COPY CopyKind
//   hdr bits 16..23: ST_LIN number of negated terms k; the accumulator starts at k * p
//   w[0]           : destination slot
//   ST_FMUL        : w[1] = a, w[2] = b
//   ST_DOT2        : w[1] = a, w[2] = b, w[3] = c, w[4] = d
//   ST_LIN         : w[1+i] = slot | (negate << 15), i < term count <= 13
//   ST_COPY        : w[1] = source slot (LDS slot, global slot or constant id)
struct OpDesc {
    uint32_t hdr;
    uint16_t w[14];
};
static_assert(sizeof(OpDesc) == 32, "OpDesc must be 32 bytes");

enum { LIN_MAX_TERMS = 13 };

// Constant table ids (each entry 12 x u32, Montgomery form).
enum ConstId : uint32_t {
    C_ONE = 0,
    C_TWO_INV = 1,
    C_FROB1_BASE = 2,   // + 2*(i-1) + {0: c0, 1: c1}, i = 1..5
    C_FROB2_BASE = 12,  // + (i-1), i = 1..5   (gamma_2^i lies in Fp)
    C_R2 = 17,          // R^2 mod p: multiplying by it converts canonical -> Montgomery
    C_RAW_ONE = 18,     // the integer 1: multiplying by it converts Montgomery -> canonical
    C_COUNT = 19,
};

}  // namespace bls381
