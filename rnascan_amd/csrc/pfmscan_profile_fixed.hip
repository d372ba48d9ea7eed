// pfmscan_profile_fixed.hip -- k_profile (codes + averaged-structure profile: all scores and the fused hits pass) with the
// PFM width as a COMPILE-TIME constant.  Same tile, same stager, same operation order and therefore the same bits as the
// width-generic kernel in pfmscan_kernels.hip (rnascan.py:302-307 for the structure rows, _pwm.c:34-68 for the letters);
// what the constant buys is VALU issue slots, the unit the headline kernel is shortest of once its bytes are moving:
//   * the row loop is straight-line code: no round counter, no per-round advance of the V letter addresses and of the row
//     pointer (27 VALU instructions per wave and tile), every LDS offset an immediate of its ds_read;
//   * the slide-in after the LAST row -- a row no window of the thread uses -- is not loaded or converted (7 + 4);
//   * the piece counts of the stager and the layout of the output staging are constants.
// Widths without an instantiation (below 9 rows by measurement, above 18) and register staging run the generic kernel
// (launch_profile_fixed says no).
#include <cstdlib>
#include "pfmscan_profile.hpp"

namespace pfmscan {

// LDS address of letter `code` (bits 0..2) in row 0 of the [m][8] fp64 table at `tbase`: v_and + v_lshl_add (hipcc's own choice is
// shift, mask, and an add of tbase + j * 64 at EVERY look-up; as an opaque value the address keeps j * 64 as the ds_read's immediate)
__device__ __forceinline__ uint32_t letter_addr(uint32_t code, uint32_t tbase)
{
    uint32_t a;
    asm("v_and_b32 %0, 7, %1\n\tv_lshl_add_u32 %0, %0, 3, %2" : "=&v"(a) : "v"(code), "s"(tbase));
    return a;
}

template <int V, bool HAS_SEQ>
__device__ __forceinline__ void pin_sums(double (&st)[V], double (&sq)[V])
{
    static_assert(V == 5, "one operand list per V");
    if (HAS_SEQ)
        asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4]), "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]), "+v"(sq[4])::"memory");
    else
        asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]), "+v"(st[4])::"memory");
}

template <int V, int MW, bool HAS_SEQ, typename PROF_T, bool FINITE>
__device__ __forceinline__ void compute_tile_fixed(const PROF_T *prof_lds, const unsigned char *code_lds, const char *tseq_lds,
                                                   const double *__restrict__ pssm, int la, double (&acc_st)[V], double (&acc_sq)[V])
{
    double rows[V][7];
    uint32_t sadr[V];                // LDS address of this slot's letter in table row 0; row j is the immediate offset j * 64
    const uint32_t tbase = lds_addr(tseq_lds);
    const PROF_T *mine = prof_lds + la * 7;           // the thread's first row: every later row is an immediate offset
    const unsigned char *cmine = code_lds + la;
#pragma unroll
    for (int s = 0; s < V; ++s) {
#pragma unroll
        for (int k = 0; k < 7; ++k) rows[s][k] = (double)mine[s * 7 + k];
        sadr[s] = HAS_SEQ ? letter_addr(cmine[s], tbase) : 0u;
        acc_st[s] = 0.0;
        acc_sq[s] = 0.0;
    }
    const __attribute__((address_space(4))) double *ptab = (const __attribute__((address_space(4))) double *)pssm;
    double Pn[7];                                     // PSSM row of the NEXT step (SGPRs), requested one step ahead
#pragma unroll
    for (int k = 0; k < 7; ++k) Pn[k] = ptab[k];
#pragma unroll
    for (int j = 0; j < MW; ++j) {
        const int u = j % V;
        double P[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) P[k] = Pn[k];
        // (a) everything the step reads, requested before its arithmetic: the V table values, the row that slides in at the
        //     end (as stored), its letter, and the next PSSM row -- their latency runs under the 35 FMAs below
        double tv[V];
        PROF_T nr[7];
        uint32_t ncode = 0;
        if (HAS_SEQ) {
#pragma unroll
            for (int v = 0; v < V; ++v) tv[v] = *(const __attribute__((address_space(3))) double *)(uintptr_t)(sadr[(u + v) % V] + j * 64);
        }
        if (j + 1 < MW) {                             // the slide-in after the last row has no reader
#pragma unroll
            for (int k = 0; k < 7; ++k) nr[k] = mine[(j + V) * 7 + k];
            if (HAS_SEQ) ncode = cmine[j + V];
#pragma unroll
            for (int k = 0; k < 7; ++k) Pn[k] = ptab[(j + 1) * 7 + k];
        }
        __builtin_amdgcn_sched_barrier(0);
        // (b) the arithmetic, in the reference's order per window
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int slot = (u + v) % V;             // holds stream position la + v + j
            if (FINITE) {
                double sacc = acc_st[v];
#pragma unroll
                for (int k = 0; k < 7; ++k) sacc = fma(rows[slot][k], P[k], sacc);
                acc_st[v] = sacc;
            } else {
                double d = rows[slot][0] * P[0];
#pragma unroll
                for (int k = 1; k < 7; ++k) d = fma(rows[slot][k], P[k], d);
                acc_st[v] += nan_to_num(d);
            }
            if (HAS_SEQ) acc_sq[v] += tv[v];
        }
        // (c) slot u is dead: position la + j + V slides in
        if (j + 1 < MW) {
#pragma unroll
            for (int k = 0; k < 7; ++k) rows[u][k] = (double)nr[k];
            if (HAS_SEQ) sadr[u] = letter_addr(ncode, tbase);
        }
        // One row step stays one step.  The unrolled loop is a single basic block, and left alone the instruction selector
        // linearises it with the sums of later steps deferred and their operands (rows, table values) held -- 230 VGPRs spilled
        // at w = 12.  The empty asm takes every running sum as an in/out operand (no instruction is emitted), which pins the
        // step's FMAs and adds in front of it; the memory clobber does the same for the LDS reads, the fence below for the
        // machine scheduler.
        pin_sums<V, HAS_SEQ>(acc_st, acc_sq);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (FINITE) {
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (!(fabs(acc_st[v]) <= DBL_MAX)) acc_st[v] = struct_window_slow(prof_lds, la + v, pssm, MW);
    }
}

template <int V, int MW, bool HAS_SEQ, typename PROF_T, bool FINITE, bool HITS>
__global__ __launch_bounds__(BLOCK, 4) void k_profile_fixed(const ScanArgs a)
{
    using L = ProfileLayout<V, PROF_T>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t tile0 = (int64_t)blockIdx.x * L::TILE;
    constexpr int prof_bytes = L::prof_bytes(MW);
    char *tseq_lds = reinterpret_cast<char *>(smem + prof_bytes + (HAS_SEQ ? L::code_bytes(MW) : 0));
    if (a.prio) __builtin_amdgcn_s_setprio(3);        // see k_profile
    stage_tile<V, HAS_SEQ, PROF_T, 2>(a, tile0, smem, MW);
    if (HAS_SEQ)
        for (int i = threadIdx.x; i < MW * 8; i += BLOCK) reinterpret_cast<double *>(tseq_lds)[i] = a.letter_table[i];
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    dma_wait_all();
    __syncthreads();
    const int la = threadIdx.x * V;
    double acc_st[V], acc_sq[V];
    compute_tile_fixed<V, MW, HAS_SEQ, PROF_T, FINITE>(reinterpret_cast<const PROF_T *>(smem), smem + prof_bytes, tseq_lds, a.struct_pssm,
                                                       la, acc_st, acc_sq);
    if (HITS) {
        settle_near<V, PROF_T>(a, reinterpret_cast<const PROF_T *>(smem), la, acc_st);
        emit_tile<V, HAS_SEQ, true>(a, tile0, la, acc_st, acc_sq, smem);      // the fused combined filter: seq > thr && struct > thr
    }
    else
        emit_tile_wave<V, HAS_SEQ, PROF_T>(a, tile0, la, acc_st, acc_sq, smem, MW);
}

template <int MW, bool HAS_SEQ, typename PROF_T, bool FINITE, bool HITS>
static hipError_t launch_fixed_inst(const ScanArgs &a, hipStream_t stream)
{
    constexpr int V = 5;       // 7 windows per thread (1792-position tiles, 162 VGPRs, 3 workgroups per CU): 2.14-2.16 ms on C3 beside 2.07-2.15
    using L = ProfileLayout<V, PROF_T>;
    const unsigned grid = (unsigned)((a.n_pos + L::TILE - 1) / L::TILE);
    const int lds = L::total(MW, HAS_SEQ, 1);
    auto kern = k_profile_fixed<V, MW, HAS_SEQ, PROF_T, FINITE, HITS>;
    static std::atomic<uint64_t> configured{0};     // per instantiation, one bit per device
    hipError_t e = allow_full_lds(reinterpret_cast<const void *>(kern), configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, stream, a);
    return hipGetLastError();
}

template <int MW, bool HITS>
static hipError_t launch_fixed_width(const ScanArgs &a, hipStream_t stream)
{
    const bool has_seq = a.letter_table != nullptr, fin = a.struct_finite != 0;
    if (a.profile_dtype == PFMSCAN_PROFILE_F64) {
        if (has_seq) return fin ? launch_fixed_inst<MW, true, double, true, HITS>(a, stream) : launch_fixed_inst<MW, true, double, false, HITS>(a, stream);
        return fin ? launch_fixed_inst<MW, false, double, true, HITS>(a, stream) : launch_fixed_inst<MW, false, double, false, HITS>(a, stream);
    }
    if (has_seq) return fin ? launch_fixed_inst<MW, true, float, true, HITS>(a, stream) : launch_fixed_inst<MW, true, float, false, HITS>(a, stream);
    return fin ? launch_fixed_inst<MW, false, float, true, HITS>(a, stream) : launch_fixed_inst<MW, false, float, false, HITS>(a, stream);
}

// true when a fixed-width instantiation took the scan (all scores, or the fused hits pass), result in *err; false: the caller
// runs the generic kernel
bool launch_profile_fixed(const ScanArgs &a, hipStream_t stream, hipError_t *err)
{
    const bool off = std::getenv("PFMSCAN_PROFILE_GENERIC") != nullptr;      // tests and A/B runs: the width-generic kernel
    if (off || a.ablate || !a.struct_pssm || !a.profile || a.out_letters_f64) return false;
    // C3 with placed arrays, generic / fixed in ms (tools/ab_fixed.sh, three interleaved pairs each, profiles/r4/NOTES.md):
    // w = 6: 2.00 / 1.99, 8: 1.960 / 1.974, 9: 1.990 / 1.976, 10: 2.038 / 1.966, 11: 2.107 / 1.993, 12: 2.188 / 2.052,
    // 16: 2.538 / 2.328, 18: 2.707 / 2.475 -- below nine rows the generic loop's two full rounds of five cost nothing extra
    int min_w = 9;
    if (const char *v = std::getenv("PFMSCAN_PROFILE_FIXED_MIN")) min_w = std::atoi(v);
    if (a.m < min_w) return false;
    switch (a.m) {
#define FIXED_WIDTH(W) case W: *err = a.hits ? launch_fixed_width<W, true>(a, stream) : launch_fixed_width<W, false>(a, stream); return true;
    FIXED_WIDTH(4) FIXED_WIDTH(5) FIXED_WIDTH(6) FIXED_WIDTH(7) FIXED_WIDTH(8) FIXED_WIDTH(9) FIXED_WIDTH(10) FIXED_WIDTH(11)
    FIXED_WIDTH(12) FIXED_WIDTH(13) FIXED_WIDTH(14) FIXED_WIDTH(15) FIXED_WIDTH(16) FIXED_WIDTH(17) FIXED_WIDTH(18)
#undef FIXED_WIDTH
    default: return false;
    }
}

}  // namespace pfmscan
