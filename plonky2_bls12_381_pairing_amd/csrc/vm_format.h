// Encoding of a lane-group program ("tape") shared by the host-side scheduler
// and the gfx950 interpreter kernel.
//
// A program is executed by groups of K adjacent lanes; one group owns one
// pairing (or one Fq12 value).  Every value is one Fp element (14 signed
// digits of 28 bits, Montgomery factor 2^392, lazily reduced) held in an LDS
// slot private to the group.  A program is a
// list of steps; in one step every lane of the group executes at most one
// operation of the step's type.  All lanes read their operands before any lane
// writes its result, so a destination may alias any operand of the same step.
#pragma once
#include <stdint.h>

namespace bls381 {

enum StepType : uint32_t {
    ST_FMUL = 0,  // dst = a * b / 2^392              (in (-p, 2p), digits normalised)
    ST_LIN  = 1,  // dst = sum_i (+/-) t_i            (digit-wise, no carries unless a step flag asks)
    ST_COPY = 2,  // LDS <-> global slot arrays, constant table -> LDS
    ST_DOT2 = 3,  // dst = (a * b +/- c * d) / 2^392  (one Montgomery reduction, in (-p, 2p))
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
                         //         bit 8: results get a carry sweep (digits back below 2^28)
                         //         bit 9: results are also brought into (-0.2p, 1.2p): quotient
                         //                estimate from the top digit
                         // multiplier and sum steps: bit 16 some lane carries a fill, bit 17 a spill
};

// One operation descriptor per (step, lane-in-group).  32 bytes.
//   hdr bit 0      : active
//   hdr bit 1      : ST_DOT2: the op has a second product (else dst = a * b)
//   hdr bit 2      : ST_DOT2: the second product is subtracted
//   Idle lanes of multiplier and sum steps carry dst = TRASH slot and operands = ZERO slot (the
//   program's last two LDS slots), so those steps run without per-lane branches.
//   hdr bits 8..15 : ST_LIN term count / ST_COPY CopyKind
//   hdr bits 24..31: ST_LIN number of negated terms
//   w[0]           : destination slot
//   ST_FMUL        : w[1] = a, w[2] = b
//   ST_DOT2        : w[1] = a, w[2] = b, w[3] = c, w[4] = d
//   ST_LIN         : w[1+i] = slot | (negate << 15), i < term count <= 7
//   ST_COPY        : w[1] = source slot (LDS slot, global slot or constant id)
//   Multiplier and sum steps may carry copies besides their operation:
//   w[9], w[10]    : fill: w[9] = LDS slot | 0x8000 (valid), w[10] = source slot | CopyKind << 12
//                    (CK_IN, CK_FILL or CK_CONST); the loads are issued as the step starts, the LDS
//                    slot is written after the step's results
//   w[11], w[12]   : spill: w[11] = LDS slot | 0x8000 (valid), w[12] = spill-array slot; read and
//                    stored as the step starts, before the step's fills are issued
struct OpDesc {
    uint32_t hdr;
    uint16_t w[14];
};
static_assert(sizeof(OpDesc) == 32, "OpDesc must be 32 bytes");

enum { LIN_MAX_TERMS = 7, MAX_LDS_SLOTS = 128 };  // seven 2^28-bounded digits and their carries fit 32 bits

// ---- device encoding.  The interpreter reads 16-byte descriptors, 64 per 1 KiB block: a block
// holds floor(64 / K) consecutive steps, the descriptor of (step j of the block, lane-in-group r)
// sits at index j * K + r, so one 16-byte load per lane fetches a whole block for a wave and the
// lanes then pick their descriptors out of each other's registers.  The step-wide fields are
// replicated into every descriptor of the step:
//   w[0] bit 0 active, bit 1 second product, bit 2 subtract it, bits 3..4 StepType,
//        bits 5..7 largest term count of the step, bits 8..9 sweep / reduce flags of the step,
//        bits 10..12 CopyKind, bits 13..15 negated-term count, bits 16..31 destination slot
//        (multiplier and sum steps: destination in bits 16..22, bit 23 some lane of the step
//        carries a fill, bit 24 some lane carries a spill)
//   w[1], w[2]  multiplier and sum steps: operand i in byte i, LDS slot in bits 0..6, negate in bit 7;
//               copy steps: w[1] = source slot
//   w[3]        carried copies: bits 0..6 fill LDS slot, 7..8 CopyKind, 9..18 source slot, 19 valid;
//               bits 20..26 spill LDS slot, 27 valid; the spill-array slot sits in w[0] bits 25..31
struct PackedDesc {
    uint32_t w[4];
};
inline PackedDesc pack_desc(const StepHdr& h, const OpDesc& d) {
    PackedDesc p{{0, 0, 0, 0}};
    const uint32_t nterms_step = h.max_terms & 0xffu, flags = (h.max_terms >> 8) & 3u;
    p.w[0] = (d.hdr & 7u) | (h.type << 3) | (nterms_step << 5) | (flags << 8) | ((uint32_t)d.w[0] << 16);
    if (h.type == ST_COPY) {
        p.w[0] |= ((d.hdr >> 8) & 7u) << 10;
        p.w[1] = d.w[1];
    } else {
        p.w[0] |= ((d.hdr >> 24) & 7u) << 13;
        p.w[0] |= ((h.max_terms >> 16) & 3u) << 23;
        for (int i = 0; i < 8; i++) {
            const uint32_t b = (d.w[1 + i] & 0x7fu) | ((d.w[1 + i] & 0x8000u) ? 0x80u : 0u);
            p.w[1 + i / 4] |= b << (8 * (i % 4));
        }
        if (d.w[9] & 0x8000u)
            p.w[3] |= (d.w[9] & 0x7fu) | (((d.w[10] >> 12) & 3u) << 7) | ((d.w[10] & 0x3ffu) << 9) | (1u << 19);
        if (d.w[11] & 0x8000u) {
            p.w[3] |= ((d.w[11] & 0x7fu) << 20) | (1u << 27);
            p.w[0] |= (d.w[12] & 0x7fu) << 25;
        }
    }
    return p;
}

// Constant table ids (each entry 14 digits of 28 bits; Montgomery factor 2^392 unless noted).
enum ConstId : uint32_t {
    C_ONE = 0,
    C_TWO_INV = 1,
    C_FROB1_BASE = 2,   // + 2*(i-1) + {0: c0, 1: c1}, i = 1..5
    C_FROB2_BASE = 12,  // + (i-1), i = 1..5   (gamma_2^i lies in Fp)
    C_IN = 17,          // the integer 2^400 mod p: x * 2^384 -> x * 2^392
    C_OUT = 18,         // the integer 2^384 mod p: x * 2^392 -> x * 2^384
    C_IN_CANON = 19,    // the integer 2^784 mod p: x -> x * 2^392
    C_OUT_CANON = 20,   // the integer 1: x * 2^392 -> x
    C_COUNT = 21,
};

}  // namespace bls381
