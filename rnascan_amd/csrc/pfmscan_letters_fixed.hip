// pfmscan_letters_fixed.hip -- k_letters (letter table only, all float32 scores: BASELINE config 2, `rnascan -p pfm seqs.fa` at
// -m ' -inf', pfmscan_pwm_calculate) with the PFM width as a COMPILE-TIME constant.  Same sums in the same order and therefore
// the same bits as the width-generic kernel (_pwm.c:34-68: score = 0.0 (double); score += M[j][col] per position; (float)score;
// NaN when a covered letter is foreign); what the constant buys is LDS cycles, the unit k_letters is shortest of:
//
//   k_letters at w = 8 spends, per 512 windows and CU, 128 LDS cycles on its 8 table look-ups per window (ds_read_b64: 2
//   cycles per wave-instruction), 34 on the wave-private transpose that turns 8 windows per lane into 16-byte stores, 10 on
//   reading its codes back from the staged tile -- 172 against ~104 cycles of VALU work: at the 1.6-1.7 GHz the chip holds
//   under this load that IS its 0.27 ms (0.69 of the HBM peak).  Here
//   * a lane owns FOUR consecutive windows per round: its four float32 scores are one 16-byte store, a wave-instruction writes
//     1 KiB contiguous with no transpose at all (-34);
//   * rows 0 and 1 are ONE look-up in a 64-entry table of pair sums built in the reference's order, p = (0.0 + T[0][c0]) + T[1][c1]
//     (the first addition of a window's sum is exact, so starting from p changes no bit) (-16);
//   * the width is a constant: exactly M + 3 table offsets are extracted per round (the generic kernel extracts W + 15 and
//     guards every row), every row offset is an immediate of its ds_read_b64.
// All code loads of the tile are issued before anything else and parked in LDS (CodeStage), then the rounds only store: no
// load ever queues behind a store (vmcnt is one in-order queue).  Widths 2 .. 32 (the reference's example PFMs are 18 wide); others run the
// generic kernel.
#include <cstdlib>
#include "pfmscan_device.hpp"

namespace pfmscan {

template <int M>
__global__ __launch_bounds__(BLOCK) void k_letters_fixed(const ScanArgs a)
{
    constexpr int W = 4;                              // windows per lane and round = one 16-byte store
    constexpr int ROUNDS = 4;
    constexpr int LET_TILE = BLOCK * W * ROUNDS;      // 4096 windows per workgroup
    constexpr int NB = M + W - 1;                     // code bytes a lane reads per round
    constexpr int NW = (NB + 3) / 4;                  // ... as dwords (the lane's first window starts on a 4-byte boundary)
    static_assert(M >= 2 && M <= 32 && NB + 1 <= CODE_HALO, "width outside this kernel's range");
    __shared__ __align__(16) double pair[64];         // [c0 + 8 c1] = (0.0 + T[0][c0]) + T[1][c1]
    __shared__ __align__(16) double tbl[M * 8];       // rows 2 .. M - 1 are used
    __shared__ __align__(16) uint8_t cbuf[LET_TILE + CODE_HALO];
    const int64_t n_pos = a.n_pos;
    const int64_t tile0 = (int64_t)blockIdx.x * LET_TILE;
    CodeStage<LET_TILE> cs;
    cs.fetch(a.codes, tile0, n_pos);
    for (int i = threadIdx.x; i < M * 8; i += BLOCK) tbl[i] = a.letter_table[i];
    if (threadIdx.x < 64) {
        double p = 0.0;                               // the reference's order: 0.0 + row 0, then + row 1
        p += a.letter_table[threadIdx.x & 7];
        p += a.letter_table[8 + (threadIdx.x >> 3)];
        pair[threadIdx.x] = p;
    }
    cs.park(cbuf);
    __syncthreads();

    // volatile: ONE ds_read_b64 per look-up.  Left alone hipcc pairs two look-ups that share an address register (row j of
    // window v and row j + 1 of window v - 1) into ds_read2_b64, which moves its 16 bytes per lane at HALF the LDS rate
    // (MI355X_MICROARCH.md, LDS table: 8 cycles per wave-instruction against 2 + 2): 80 of this kernel's 112 look-ups per
    // round went that way and the LDS array was busy for the kernel's whole 0.275 ms (SQ_LDS_IDX_ACTIVE, profiles/r5)
    typedef const volatile __attribute__((address_space(3))) double *lds_f64;
    typedef const __attribute__((address_space(3))) char *lds_bytes;
    const lds_bytes tb = (lds_bytes)reinterpret_cast<const char *>(tbl);
    const lds_bytes pb = (lds_bytes)reinterpret_cast<const char *>(pair);
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int local = it * (BLOCK * W) + (int)threadIdx.x * W;
        uint32_t w[NW];
#pragma unroll
        for (int d = 0; d < NW; ++d) w[d] = (*reinterpret_cast<const uint32_t *>(cbuf + local + 4 * d) & 0x07070707u) << 3;   // byte = code * sizeof(double)
        uint32_t adr[NB];
        static_for<0, NB>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            adr[q] = (w[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
        });
        double acc[W];
        static_for<0, W>([&](auto vc) __attribute__((always_inline)) {
            constexpr int v = decltype(vc)::value;
            acc[v] = *(lds_f64)(pb + (adr[v] + (adr[v + 1] << 3)));
        });
        static_for<2, M>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            static_for<0, W>([&](auto vc) __attribute__((always_inline)) {
                constexpr int v = decltype(vc)::value;
                acc[v] += *(lds_f64)(tb + j * 64 + adr[j + v]);
            });
        });
        const int64_t p0 = tile0 + local;
        float *o = a.out_seq + p0;
        const f32x4 r = {(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
        if (p0 + 4 <= n_pos) {
            __builtin_nontemporal_store(r, reinterpret_cast<f32x4 *>(o));
        } else {
            for (int v = 0; v < 4; ++v)
                if (p0 + v < n_pos) o[v] = r[v];
        }
    }
}

// true when a fixed-width instantiation took the scan (all float32 scores of a letters-only motif), result in *err
bool launch_letters_fixed(const ScanArgs &a, hipStream_t stream, hipError_t *err)
{
    if (a.hits || a.struct_pssm || !a.letter_table || !a.codes || !a.out_seq || a.out_letters_f64 || a.ablate) return false;
    if (std::getenv("PFMSCAN_LETTERS_GENERIC")) return false;                  // tests and A/B runs: the width-generic kernel
    constexpr int LET_TILE = BLOCK * 16;
    const unsigned grid = (unsigned)((a.n_pos + LET_TILE - 1) / LET_TILE);
    switch (a.m) {
#define FIXED_WIDTH(W) case W: hipLaunchKernelGGL((k_letters_fixed<W>), dim3(grid), dim3(BLOCK), 0, stream, a); break;
    FIXED_WIDTH(2) FIXED_WIDTH(3) FIXED_WIDTH(4) FIXED_WIDTH(5) FIXED_WIDTH(6) FIXED_WIDTH(7) FIXED_WIDTH(8) FIXED_WIDTH(9)
    FIXED_WIDTH(10) FIXED_WIDTH(11) FIXED_WIDTH(12) FIXED_WIDTH(13) FIXED_WIDTH(14) FIXED_WIDTH(15) FIXED_WIDTH(16)
    FIXED_WIDTH(17) FIXED_WIDTH(18) FIXED_WIDTH(19) FIXED_WIDTH(20) FIXED_WIDTH(21) FIXED_WIDTH(22) FIXED_WIDTH(23) FIXED_WIDTH(24)
    FIXED_WIDTH(25) FIXED_WIDTH(26) FIXED_WIDTH(27) FIXED_WIDTH(28) FIXED_WIDTH(29) FIXED_WIDTH(30) FIXED_WIDTH(31) FIXED_WIDTH(32)
#undef FIXED_WIDTH
    default: return false;
    }
    *err = hipGetLastError();
    return true;
}

}  // namespace pfmscan
