// Round 6, second hypothesis for the finding of DESIGN.md 3.3: does a VALU instruction that OVERWRITES the address register of a
// global load issued a few instructions earlier change the address the load uses for its last lanes when the CU's vector-memory
// path is congested by another kernel?  (limb_assign_kernel's SLP build re-used the offset registers of its map loads 3-10
// instructions behind their issue; the build that never failed kept them ~90 instructions.)
// Victim: per round 8 back-to-back  global_load_dword d_j, a_j, s[base]  from a table whose word i holds i, then GAP wait states,
// then  v_add_u32 a_j, a_j, delta  (delta points into a second table whose word i holds 0xBAD00000 | i), then s_waitcnt vmcnt(0)
// and the check d_j == index.  A load that returns a 0xBAD... word read its address after the overwrite.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/exp/vmem_war_victim.hip -o tools/exp/vmem_war_victim.so
#include <hip/hip_runtime.h>

template <int GAP>
__global__ __launch_bounds__(256) void war_victim(const unsigned* __restrict__ table, unsigned words, unsigned delta_bytes,
                                                  int active, int rounds, unsigned* __restrict__ hist, unsigned* __restrict__ detail) {
  const int tid = threadIdx.x;
  if (tid >= active) return;
  const unsigned long long base = reinterpret_cast<unsigned long long>(table);
  unsigned idx = (blockIdx.x * 256u + tid) * 2654435761u;
  for (int r = 0; r < rounds; ++r) {
    unsigned a[8], want[8], d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      idx = idx * 1664525u + 1013904223u;
      want[j] = idx % words;
      a[j] = want[j] * 4u;
    }
    asm volatile(
        "global_load_dword %0, %8, %16\n"
        "global_load_dword %1, %9, %16\n"
        "global_load_dword %2, %10, %16\n"
        "global_load_dword %3, %11, %16\n"
        "global_load_dword %4, %12, %16\n"
        "global_load_dword %5, %13, %16\n"
        "global_load_dword %6, %14, %16\n"
        "global_load_dword %7, %15, %16\n"
        ".rept %18\n s_nop 0\n .endr\n"
        "v_add_u32 %8, %8, %17\n"
        "v_add_u32 %9, %9, %17\n"
        "v_add_u32 %10, %10, %17\n"
        "v_add_u32 %11, %11, %17\n"
        "v_add_u32 %12, %12, %17\n"
        "v_add_u32 %13, %13, %17\n"
        "v_add_u32 %14, %14, %17\n"
        "v_add_u32 %15, %15, %17\n"
        "s_waitcnt vmcnt(0)\n"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+v"(a[0]),
          "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "s"(base), "v"(delta_bytes), "n"(GAP)
        : "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (d[j] != want[j]) {
        atomicAdd(&hist[tid & 63], 1u);
        const unsigned k = atomicAdd(&hist[64], 1u);
        if (k < 64) {
          detail[4 * k + 0] = blockIdx.x * 256u + tid;
          detail[4 * k + 1] = j | (r << 8);
          detail[4 * k + 2] = d[j];
          detail[4 * k + 3] = want[j];
        }
      }
      idx += a[j] & 1u;  // (keeps the overwritten registers alive: always 0)
    }
  }
}

extern "C" int war_victim_launch(int gap, int blocks, int active, int rounds, const void* table, unsigned words,
                                 unsigned delta_bytes, void* hist, void* detail, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned* t = static_cast<const unsigned*>(table);
  unsigned* h = static_cast<unsigned*>(hist);
  unsigned* d = static_cast<unsigned*>(detail);
  switch (gap) {
    case 0: hipLaunchKernelGGL(war_victim<0>, dim3(blocks), dim3(256), 0, s, t, words, delta_bytes, active, rounds, h, d); break;
    case 2: hipLaunchKernelGGL(war_victim<2>, dim3(blocks), dim3(256), 0, s, t, words, delta_bytes, active, rounds, h, d); break;
    case 8: hipLaunchKernelGGL(war_victim<8>, dim3(blocks), dim3(256), 0, s, t, words, delta_bytes, active, rounds, h, d); break;
    default: hipLaunchKernelGGL(war_victim<32>, dim3(blocks), dim3(256), 0, s, t, words, delta_bytes, active, rounds, h, d); break;
  }
  return (int)hipGetLastError();
}
