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
    ST_FMUL = 0,  // dst = a * b / 2^392              (in (-p, 2p), digits normalised); a, b: signed sums of up to 3 slots
    ST_LIN  = 1,  // dst = sum_i k_i * t_i            (digit-wise, small integer k_i, no carries unless a step flag asks)
    ST_COPY = 2,  // LDS <-> global slot arrays, constant table -> LDS
    ST_DOT2 = 3,  // dst = (a * b + c * d) / 2^392    (one Montgomery reduction, in (-p, 2p)); operands as for ST_FMUL
    ST_DOTN = 4,  // dst = (sum of +-a_t * b_t) / 2^392, t < n <= 8 (one reduction); every operand one slot
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
    uint32_t max_terms;  // bits 0..7: ST_LIN: largest term count among the step's ops (<= 7);
                         //            multiplier steps: largest number of terms in any operand (1..3)
                         //            ST_DOTN: largest number of products among the step's ops (3..8)
                         // ST_LIN: bit 8: results get a carry sweep (digits back below 2^28)
                         //         bit 9: results are also brought into (-0.2p, 1.2p): quotient
                         //                estimate from the top digit
                         // multiplier and sum steps: bit 16 some lane carries a fill, bit 17 a spill
};

// One operation descriptor per (step, lane-in-group).  48 bytes.
//   hdr bit 0      : active
//   hdr bit 1      : ST_DOT2: the op has a second product (else dst = a * b)
//   Idle lanes of multiplier and sum steps carry dst = TRASH slot and operands = ZERO slot (the
//   program's last two LDS slots), so those steps run without per-lane branches.
//   hdr bits 8..15 : ST_LIN term count / ST_COPY CopyKind
//   hdr bits 24..27: ST_LIN: the sum starts from a shifted zero: every digit moves by B * 2^28, B this signed 4-bit
//                    number (the digit vector added has the value 0); the step sweeps
//   w[0]           : destination slot
//   ST_FMUL, ST_DOT2: every operand is a signed sum of up to three slots; term t of operand j (a, b, c, d = 0..3)
//                    is w[1 + 3 j + t] = slot | (negate << 15), missing terms are the ZERO slot.  The result is
//                    a * b (+ c * d): a difference of products is expressed through the signs of d's terms.  In a
//                    step whose header says one term per operand only d's sign is honoured.
//   ST_DOTN        : product t is w[1 + 2 t] * w[2 + 2 t] for t < 6 and w[17 + 2 (t - 6)] * w[18 + 2 (t - 6)] for t = 6, 7
//                    (the words between hold the carried copies): slot, and on the second factor negate << 15;
//                    missing products are ZERO * ZERO
//   ST_LIN         : w[1 + i] = slot | ((coefficient & 0xff) << 7), i < term count <= 7; coefficients are
//                    signed 8-bit integers, never zero
//   ST_COPY        : w[1] = source slot (LDS slot, global slot or constant id)
//   Multiplier and sum steps may carry copies besides their operation:
//   w[13], w[14]   : fill: w[13] = LDS slot | 0x8000 (valid), w[14] = source slot | CopyKind << 12
//                    (CK_IN, CK_FILL or CK_CONST); the loads are issued as the step starts, the LDS
//                    slot is written after the step's results
//   w[15], w[16]   : spill: w[15] = LDS slot | 0x8000 (valid), w[16] = spill-array slot; read and
//                    stored as the step starts, before the step's fills are issued
struct OpDesc {
    uint32_t hdr;
    uint16_t w[22];
};
static_assert(sizeof(OpDesc) == 48, "OpDesc must be 48 bytes");
enum { OPDESC_U32 = 12, DW_FILL = 13, DW_SPILL = 15, DW_DOTN_HI = 17 };

enum { LIN_MAX_TERMS = 7, MUL_MAX_TERMS = 3, DOTN_MAX_PAIRS = 8, LIN_MAX_COEF = 127, MAX_LDS_SLOTS = 128 };

// ---- device encoding.  The interpreter reads 32-byte descriptors, 64 per 2 KiB block, as two planes of
// 64 x 16 bytes (words 0..3 of every descriptor, then words 4..7): a block holds floor(64 / K)
// consecutive steps, the descriptor of (step j of the block, lane-in-group r) sits at index j * K + r,
// so two 16-byte loads per lane fetch a whole block for a wave and the lanes then pick their
// descriptors out of each other's registers.  The step-wide fields are replicated into every
// descriptor of the step:
//   w[0] bit 0 active, bit 1 second product, bits 2..4 StepType,
//        bits 5..7 the step's largest term count (sums: terms of an op; multiplier steps: terms of an operand;
//        ST_DOTN: products of an op, minus one; copy steps: the CopyKind), bits 8..9 sweep / reduce flags of the step,
//        bit 10 the descriptor also holds the sum of the step that followed (see pack_second_sum),
//        sums: bits 12..15 the shifted zero B (signed),
//        multiplier and sum steps: destination slot in bits 16..22, bit 23 some lane of the step
//        carries a fill, bit 24 some lane carries a spill; copy steps: bits 16..31 zero
//   multiplier steps: w[1 + t], t = 0..2: term t of the operands a, b, c, d in bytes 0..3: LDS slot in bits 0..6,
//        negate in bit 7
//   ST_DOTN: w[1 + t / 2], half t % 2: product t: first factor's slot in bits 0..6, second factor's in bits 8..14,
//        negate in bit 15
//   sum steps: w[1 + i / 2], half i % 2: term i: LDS slot in bits 0..6, coefficient in bits 7..14
//   copy steps: w[1] = source slot, w[2] = destination slot
//   w[5] carried copies: bits 0..6 fill LDS slot, 7..8 CopyKind, 9..18 source slot, 19 valid;
//        bits 20..26 spill LDS slot, 27 valid; the spill-array slot sits in w[0] bits 25..31
struct PackedDesc {
    uint32_t w[8];
};
inline PackedDesc pack_desc(const StepHdr& h, const OpDesc& d) {
    PackedDesc p{{0, 0, 0, 0, 0, 0, 0, 0}};
    const uint32_t nterms_step = h.max_terms & 0xffu, flags = (h.max_terms >> 8) & 3u;
    p.w[0] = (d.hdr & 3u) | ((h.type & 7u) << 2) | ((h.type == ST_DOTN ? nterms_step - 1 : nterms_step) << 5) | (flags << 8);
    if (h.type == ST_COPY) {
        p.w[0] |= ((d.hdr >> 8) & 7u) << 5;  // (the term-count field is the copy kind here)
        p.w[1] = d.w[1];
        p.w[2] = d.w[0];  // (a copy's destination may be any slot of a global array: a field of its own)
    } else {
        p.w[0] |= (uint32_t)d.w[0] << 16;
        p.w[0] |= ((h.max_terms >> 16) & 3u) << 23;
        if (h.type == ST_LIN) {
            const uint32_t B = (d.hdr >> 24) & 0xfu;
            p.w[0] |= B << 12;
            for (int i = 0; i < LIN_MAX_TERMS; i++) p.w[1 + i / 2] |= (uint32_t)(d.w[1 + i] & 0x7fffu) << (16 * (i % 2));
        } else if (h.type == ST_DOTN) {
            for (int t = 0; t < DOTN_MAX_PAIRS; t++) {
                const uint16_t a = d.w[t < 6 ? 1 + 2 * t : DW_DOTN_HI + 2 * (t - 6)], b = d.w[t < 6 ? 2 + 2 * t : DW_DOTN_HI + 1 + 2 * (t - 6)];
                p.w[1 + t / 2] |= (uint32_t)((a & 0x7fu) | ((b & 0x7fu) << 8) | ((b & 0x8000u) ? 0x8000u : 0u)) << (16 * (t % 2));
            }
        } else {
            for (int j = 0; j < 4; j++)
                for (int t = 0; t < MUL_MAX_TERMS; t++) {
                    const uint16_t e = d.w[1 + 3 * j + t];
                    p.w[1 + t] |= (uint32_t)((e & 0x7fu) | ((e & 0x8000u) ? 0x80u : 0u)) << (8 * j);
                }
        }
        if (d.w[DW_FILL] & 0x8000u)
            p.w[5] |= (d.w[DW_FILL] & 0x7fu) | (((d.w[DW_FILL + 1] >> 12) & 3u) << 7) | ((d.w[DW_FILL + 1] & 0x3ffu) << 9) | (1u << 19);
        if (d.w[DW_SPILL] & 0x8000u) {
            p.w[5] |= ((d.w[DW_SPILL] & 0x7fu) << 20) | (1u << 27);
            p.w[0] |= (d.w[DW_SPILL + 1] & 0x7fu) << 25;
        }
    }
    return p;
}

// A sum step of at most four terms per sum that carries no copies can share the descriptors, and so the dispatch,
// of the step before it (a multiplier step, or a sum step of at most six terms): the kernel runs the first step's
// operation, then the second step's sum, in program order.  The second sum sits in the words the first leaves free:
//   w[4] bits 0..6 destination slot, 8..11 its shifted zero B, 12..14 the second step's largest term count,
//        15..16 the second step's sweep / reduce flags
//   w[6], w[7] its terms, as w[1], w[2] of a sum step
// (Measured with sums of up to six terms as well, the fifth and sixth in w[3] where the first step leaves it free:
// more pairs, but the longer sum routine in that place costs more than they save.)
inline bool can_take_second_sum(const StepHdr& first, const StepHdr& second) {
    const bool first_ok = first.type == ST_FMUL || first.type == ST_DOT2 || (first.type == ST_LIN && (first.max_terms & 0xffu) <= 6);
    return first_ok && second.type == ST_LIN && (second.max_terms & 0xffu) <= 4 && ((second.max_terms >> 16) & 3u) == 0;
}
inline void pack_second_sum(PackedDesc& p, const StepHdr& h, const OpDesc& d) {
    p.w[0] |= 1u << 10;
    p.w[4] = (uint32_t)(d.w[0] & 0x7fu) | (((d.hdr >> 24) & 0xfu) << 8) | ((h.max_terms & 7u) << 12) | (((h.max_terms >> 8) & 3u) << 15);
    p.w[6] = p.w[7] = 0;
    for (int i = 0; i < 4; i++) p.w[6 + i / 2] |= (uint32_t)(d.w[1 + i] & 0x7fffu) << (16 * (i % 2));
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
