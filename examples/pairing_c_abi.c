/* Minimal use of the C ABI from plain C: one pairing of the generators, printed as hex limbs.
 *   gcc -std=c99 -I include examples/pairing_c_abi.c -L plonky2_bls12_381_pairing_amd -lbls381_hip \
 *       -Wl,-rpath,$PWD/plonky2_bls12_381_pairing_amd -o /tmp/pairing_c_abi && /tmp/pairing_c_abi
 * Inputs and outputs are the reference's own representation: Fq = 6 little-endian u64 in Montgomery form
 * (ark-ff BigInt<6>), G1 = {x, y}, G2 = {x.c0, x.c1, y.c0, y.c1}, Fq12 = c0.c0.c0 ... c1.c2.c1. */
#include <inttypes.h>
#include <stdio.h>

#include "bls381_hip.h"

/* the standard generators, Montgomery form */
static const uint64_t G1_GEN[12] = {
    0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull,
    0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
static const uint64_t G2_GEN[24] = {
    0xf5f28fa202940a10ull, 0xb3f5fb2687b4961aull, 0xa1a893b53e2ae580ull, 0x9894999d1a3caee9ull, 0x6f67b7631863366bull, 0x058191924350bcd7ull,
    0xa5a9c0759e23f606ull, 0xaaa0c59dbccd60c3ull, 0x3bb17e18e2867806ull, 0x1b1ab6cc8541b367ull, 0xc2b6ed0ef2158547ull, 0x11922a097360edf3ull,
    0x4c730af860494c4aull, 0x597cfa1f5e369c5aull, 0xe7e6856caa0a635aull, 0xbbefb5e96e0d495full, 0x07d3a975f0ef25a2ull, 0x0083fd8e7e80dae5ull,
    0xadc0fc92df64b05dull, 0x18aa270a2b1461dcull, 0x86adac6a3be4eba0ull, 0x79495c4ec93da33aull, 0xe7175850a43ccaedull, 0x0b2bc2a163de1bf2ull};

int main(void) {
    bls381_ctx* ctx = NULL;
    if (bls381_ctx_create(&ctx, 0) != BLS381_OK) {
        fprintf(stderr, "bls381_ctx_create: %s\n", bls381_last_error());
        return 1;
    }
    uint64_t gt[72];
    if (bls381_pairing_batch(ctx, G1_GEN, G2_GEN, NULL, 1, gt) != BLS381_OK) {
        fprintf(stderr, "bls381_pairing_batch: %s\n", bls381_last_error());
        bls381_ctx_destroy(ctx);
        return 1;
    }
    for (int k = 0; k < 12; k++) {
        printf("c%d:", k);
        for (int l = 5; l >= 0; l--) printf(" %016" PRIx64, gt[6 * k + l]);
        printf("\n");
    }
    bls381_ctx_destroy(ctx);
    return 0;
}
