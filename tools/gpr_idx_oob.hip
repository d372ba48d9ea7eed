// gpr_idx_oob.hip -- what went wrong in k_letters_cred8<16> (profiles/r5/NOTES.md), isolated.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gpr_idx_oob.hip -o tools/gpr_idx_oob && tools/gpr_idx_oob      (-O1: correct)
//
// A private array that is indexed by a value the compiler does not know (a loop it left rolled) is promoted to a vector
// register tuple; the GUARDED update `if (0 <= u && u <= 16) pk[u] += x` is if-converted into an UNCONDITIONAL indexed
// register write (s_set_gpr_idx_on u, gpr_idx(DST); v_mov_b32 v[base], x) followed by a select on the guard.  The index is
// not clamped: for u outside the array the write lands on v[base + u], whatever lives there.  Here `keep[]` lives there.
// Prints the number of threads whose keep[] values came back changed (0 = the pattern is harmless in this build).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void k(const uint32_t *add, const uint32_t *seed, uint32_t *out, int n)
{
    uint32_t pk[17], keep[40];
    for (int i = 0; i < 17; ++i) pk[i] = 0u;
    for (int i = 0; i < 40; ++i) {
        keep[i] = seed[i] ^ threadIdx.x;
        asm volatile("" : "+v"(keep[i]));          // opaque to the optimiser and live across the loop, in VGPRs
    }
    for (int q = 0; q < n; ++q) {                  // trip count unknown: stays a loop, pk[] is indexed at run time
        uint32_t dj[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) dj[k] = add[16 * q + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {             // the kernel's update: row pair k of position q belongs to window q - 2k
            const int u = q - 2 * k;
            if (u >= 0 && u <= 16) pk[u] += dj[k]; // a cheap guarded block: if-converted
        }
    }
    uint32_t s = 0, bad = 0;
    for (int i = 0; i < 17; ++i) s += pk[i];
    for (int i = 0; i < 40; ++i) {
        asm volatile("" : "+v"(keep[i]));
        bad += keep[i] != (seed[i] ^ threadIdx.x);
    }
    out[2 * (blockIdx.x * 256 + threadIdx.x)] = s;
    out[2 * (blockIdx.x * 256 + threadIdx.x) + 1] = bad;
}

int main()
{
    const int n = 47;                              // q = 0 .. 46 as for PFMs 32 wide: u = q - 2k runs over -30 .. 46
    std::vector<uint32_t> add(16 * n, 1u), seed(40);
    uint32_t want = 0;
    for (int q = 0; q < n; ++q)
        for (int k = 0; k < 16; ++k) want += q - 2 * k >= 0 && q - 2 * k <= 16;
    for (int i = 0; i < 40; ++i) seed[i] = 0x9E3779B9u * (i + 1);
    uint32_t *d_add, *d_seed, *d_out;
    if (hipMalloc(&d_add, 64 * n) || hipMalloc(&d_seed, 160) || hipMalloc(&d_out, 2048)) return 2;
    if (hipMemcpy(d_add, add.data(), 64 * n, hipMemcpyHostToDevice) || hipMemcpy(d_seed, seed.data(), 160, hipMemcpyHostToDevice)) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d_add, d_seed, d_out, n);
    std::vector<uint32_t> out(512);
    if (hipMemcpy(out.data(), d_out, 2048, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("hip error\n"); return 2; }
    int wrong_sum = 0, clobbered = 0;
    for (int t = 0; t < 256; ++t) { wrong_sum += out[2 * t] != want; clobbered += out[2 * t + 1] != 0; }
    std::printf("pk sum of thread 0: %u (want %u); threads with a wrong pk sum: %d, threads with clobbered keep[] registers: %d (of 256)\n",
                out[0], want, wrong_sum, clobbered);
    return (wrong_sum || clobbered) ? 1 : 0;
}
