// kernels.hip — the one device translation unit of libnbody_hip.so: the kernels of nbody_kernels.hpp and the launch functions that
// pick an instantiation (namespace nbl, declared in nbody_internal.hpp).  No context, no options, no policy here: WHAT to launch is
// decided in context.cpp (launch_force) and arrives as a KernelSel + ForceArgs; this file turns that into one hipLaunchKernelGGL.
// gfx950 only.
#include <hip/hip_runtime.h>

#include "nbody_internal.hpp"
#include "nbody_kernels.hpp"

using namespace nbk;

#define NBL_HIDDEN __attribute__((visibility("hidden")))

namespace {

template <typename K>
int launch_k(K kernel, const nbl::KernelSel& s, hipStream_t stream, dim3 grid, const ForceArgs& a) {
  if (s.dyn_lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kernel, grid, dim3(wg_threads(a.wsplit)), s.dyn_lds, stream, a);
  return (int)hipGetLastError();
}

template <int R, int ARITH>
int launch_f32_RA(const nbl::KernelSel& s, hipStream_t st, dim3 grid, const ForceArgs& a) {
  if constexpr (R == 8) {   // 8 bodies per lane only exists for the SMEM variant
    return launch_k(force_smem_f32<R, ARITH, 1>, s, st, grid, a);
  } else {
    switch (s.variant) {
      case NBODY_VARIANT_LDS:
        if (s.tile >= 1024) return launch_k(force_lds_f32<R, ARITH, 1024>, s, st, grid, a);
        if (s.tile >= 512) return launch_k(force_lds_f32<R, ARITH, 512>, s, st, grid, a);
        return launch_k(force_lds_f32<R, ARITH, 256>, s, st, grid, a);
      case NBODY_VARIANT_READLANE:
        return launch_k(force_readlane_f32<R, ARITH>, s, st, grid, a);
      default:
        if constexpr (R == 1) {
          if (a.wsplit == 16) return launch_k(force_smem_f32<1, ARITH, 16>, s, st, grid, a);
          if (a.wsplit == 4) return launch_k(force_smem_f32<1, ARITH, 4>, s, st, grid, a);
        }
        return launch_k(force_smem_f32<R, ARITH, 1>, s, st, grid, a);
    }
  }
}

template <int R>
int launch_f32_R(const nbl::KernelSel& s, hipStream_t st, dim3 grid, const ForceArgs& a) {
  switch (s.arith) {
    case NBODY_ARITH_REFERENCE: return launch_f32_RA<R, 1>(s, st, grid, a);
    case NBODY_ARITH_STRICT: return launch_f32_RA<R, 2>(s, st, grid, a);
    case NBODY_ARITH_REFERENCE_STRICT: return launch_f32_RA<R, 3>(s, st, grid, a);
    default: return launch_f32_RA<R, 0>(s, st, grid, a);
  }
}

// the hand-scheduled fp32 loop in form PH (NBODY_OPT_ISA_PHASE), with or without the wave split
template <int PH>
int launch_isa_f32(const nbl::KernelSel& s, hipStream_t st, dim3 grid, const ForceArgs& a) {
  if constexpr (PH <= 1) {   // the 16-wave form exists for the product loop and its placement twin (resolve_config sees to it)
    if (a.wsplit == 16) return launch_k(force_isa_f32<PH, 16>, s, st, grid, a);
  }
  return a.wsplit == 4 ? launch_k(force_isa_f32<PH, 4>, s, st, grid, a) : launch_k(force_isa_f32<PH, 1>, s, st, grid, a);
}
template <int PH>
int launch_isa_f64(const nbl::KernelSel& s, hipStream_t st, dim3 grid, const ForceArgs& a) {
  if (a.wsplit == 16) return launch_k(force_isa_f64<PH, 16>, s, st, grid, a);
  return a.wsplit == 4 ? launch_k(force_isa_f64<PH, 4>, s, st, grid, a) : launch_k(force_isa_f64<PH, 1>, s, st, grid, a);
}

}  // namespace

namespace nbl {

NBL_HIDDEN bool diag_build() {
#ifdef NBODY_DIAG_LOOPS
  return true;
#else
  return false;
#endif
}

NBL_HIDDEN int launch_force_kernel(const KernelSel& s, hipStream_t st, dim3 grid, const ForceArgs& a) {
  const int R = s.R;
  if (s.fp64 && s.variant == NBODY_VARIANT_ISA) {
    if (s.isa_phase == 2) return launch_isa_f64<2>(s, st, grid, a);
    return s.isa_phase == 0 ? launch_isa_f64<0>(s, st, grid, a) : launch_isa_f64<1>(s, st, grid, a);
  }
  if (s.fp64 && (s.arith & 2)) {   // NBODY_ARITH_STRICT / _REFERENCE_STRICT: IEEE 1/sqrt (fp64 has one d2 form: the bit-0 distinction is fp32's)
    switch (R) {
      case 1:
        if (a.wsplit == 16) return launch_k(force_smem_f64<1, 16, 1>, s, st, grid, a);
        return a.wsplit == 4 ? launch_k(force_smem_f64<1, 4, 1>, s, st, grid, a) : launch_k(force_smem_f64<1, 1, 1>, s, st, grid, a);
      case 2: return launch_k(force_smem_f64<2, 1, 1>, s, st, grid, a);
      default: return launch_k(force_smem_f64<4, 1, 1>, s, st, grid, a);
    }
  }
  if (s.fp64) {
    switch (R) {
      case 1:
        if (a.wsplit == 16) return launch_k(force_smem_f64<1, 16>, s, st, grid, a);
        return a.wsplit == 4 ? launch_k(force_smem_f64<1, 4>, s, st, grid, a) : launch_k(force_smem_f64<1, 1>, s, st, grid, a);
      case 2: return launch_k(force_smem_f64<2, 1>, s, st, grid, a);
      default: return launch_k(force_smem_f64<4, 1>, s, st, grid, a);
    }
  }
  if (a.fpga16 && s.fpga_rows16) {   // small launches of ONE segment: sixteen rows x sixteen chains per 256-thread workgroup (grid.x counts 16-row units)
    switch (s.arith) {
      case NBODY_ARITH_REFERENCE: hipLaunchKernelGGL(force_fpga16r_f32<1>, grid, dim3(256), 0, st, a); break;
      case NBODY_ARITH_STRICT: hipLaunchKernelGGL(force_fpga16r_f32<2>, grid, dim3(256), 0, st, a); break;
      case NBODY_ARITH_REFERENCE_STRICT: hipLaunchKernelGGL(force_fpga16r_f32<3>, grid, dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL(force_fpga16r_f32<0>, grid, dim3(256), 0, st, a); break;
    }
    return (int)hipGetLastError();
  }
  if (a.fpga16 && a.wsplit == 16 && s.fpga_lds) {   // sources staged through LDS (the default of the sixteen-wave form)
    switch (s.arith) {
      case NBODY_ARITH_REFERENCE: return launch_k(force_fpga16w_lds_f32<1>, s, st, grid, a);
      case NBODY_ARITH_STRICT: return launch_k(force_fpga16w_lds_f32<2>, s, st, grid, a);
      case NBODY_ARITH_REFERENCE_STRICT: return launch_k(force_fpga16w_lds_f32<3>, s, st, grid, a);
      default: return launch_k(force_fpga16w_lds_f32<0>, s, st, grid, a);
    }
  }
  if (a.fpga16 && a.wsplit == 16) {
    switch (s.arith) {
      case NBODY_ARITH_REFERENCE: return launch_k(force_fpga16w_f32<1>, s, st, grid, a);
      case NBODY_ARITH_STRICT: return launch_k(force_fpga16w_f32<2>, s, st, grid, a);
      case NBODY_ARITH_REFERENCE_STRICT: return launch_k(force_fpga16w_f32<3>, s, st, grid, a);
      default: return launch_k(force_fpga16w_f32<0>, s, st, grid, a);
    }
  }
  if (a.fpga16) {
    switch (s.arith) {
      case NBODY_ARITH_REFERENCE: return launch_k(force_fpga16_f32<1>, s, st, grid, a);
      case NBODY_ARITH_STRICT: return launch_k(force_fpga16_f32<2>, s, st, grid, a);
      case NBODY_ARITH_REFERENCE_STRICT: return launch_k(force_fpga16_f32<3>, s, st, grid, a);
      default: return launch_k(force_fpga16_f32<0>, s, st, grid, a);
    }
  }
  if (s.variant == NBODY_VARIANT_ISA) {   // one body per lane, fast arithmetic only (resolve_config guarantees both)
    if (a.long_buffers) {
      if (a.wsplit == 16) return launch_k(force_isa_long_f32<16>, s, st, grid, a);
      return a.wsplit == 4 ? launch_k(force_isa_long_f32<4>, s, st, grid, a) : launch_k(force_isa_long_f32<1>, s, st, grid, a);
    }
    switch (s.isa_phase) {
      case 0: return launch_isa_f32<0>(s, st, grid, a);
#ifdef NBODY_DIAG_LOOPS
      case 2: return launch_isa_f32<2>(s, st, grid, a);     // 2, 9..13, 16..18: the same operations in other encodings (bit-identical)
      case 3: return launch_isa_f32<3>(s, st, grid, a);     // 3..8, 14, 15: TIMING-ONLY forms, WRONG RESULTS
      case 4: return launch_isa_f32<4>(s, st, grid, a);
      case 5: return launch_isa_f32<5>(s, st, grid, a);
      case 6: return launch_isa_f32<6>(s, st, grid, a);
      case 7: return launch_isa_f32<7>(s, st, grid, a);
      case 8: return launch_isa_f32<8>(s, st, grid, a);
      case 9: return launch_isa_f32<9>(s, st, grid, a);
      case 10: return launch_isa_f32<10>(s, st, grid, a);
      case 11: return launch_isa_f32<11>(s, st, grid, a);
      case 12: return launch_isa_f32<12>(s, st, grid, a);
      case 13: return launch_isa_f32<13>(s, st, grid, a);
      case 14: return launch_isa_f32<14>(s, st, grid, a);
      case 15: return launch_isa_f32<15>(s, st, grid, a);
      case 16: return launch_isa_f32<16>(s, st, grid, a);
      case 17: return launch_isa_f32<17>(s, st, grid, a);
      case 18: return launch_isa_f32<18>(s, st, grid, a);
      case 19: return launch_isa_f32<19>(s, st, grid, a);
      case 20: return launch_isa_f32<20>(s, st, grid, a);
#endif
      default: return launch_isa_f32<1>(s, st, grid, a);
    }
  }
  switch (R) {
    case 1: return launch_f32_R<1>(s, st, grid, a);
    case 2: return launch_f32_R<2>(s, st, grid, a);
    case 8: return launch_f32_R<8>(s, st, grid, a);
    default: return launch_f32_R<4>(s, st, grid, a);
  }
}

// the two-launch form (NBODY_OPT_FUSE_COMBINE = 0): after the step's last force launch, add the partials
NBL_HIDDEN int launch_combine_kernel(int fp64, hipStream_t st, dim3 grid, const ForceArgs& c) {
  if (fp64) hipLaunchKernelGGL((combine_kernel<double, d4>), grid, dim3(kBlock), 0, st, c);
  else hipLaunchKernelGGL((combine_kernel<float, f4>), grid, dim3(kBlock), 0, st, c);
  return (int)hipGetLastError();
}

NBL_HIDDEN int launch_drift_kernel(int fp64, hipStream_t st, void* pos_rows, const void* vel, int n_rows, float dt, double dt64) {
  if (n_rows <= 0) return 0;
  dim3 grid((n_rows + kBlock - 1) / kBlock);
  if (fp64) hipLaunchKernelGGL((drift_kernel<double, d4>), grid, dim3(kBlock), 0, st, (d4*)pos_rows, (const d4*)vel, n_rows, dt, dt64);
  else hipLaunchKernelGGL((drift_kernel<float, f4>), grid, dim3(kBlock), 0, st, (f4*)pos_rows, (const f4*)vel, n_rows, dt, dt64);
  return (int)hipGetLastError();
}

// RAM A's read port: body words of the host's RAM image into the resident source array (S/top_level.vhd:206-208, 238-240); t0 (may be
// null): where its first wave stamps the start of the request's tick count
NBL_HIDDEN int launch_ingest_kernel(hipStream_t st, void* dst_words, const void* ram_a_bodies, int n, unsigned long long* t0) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(ingest_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, (f4*)dst_words, (const f4*)ram_a_bodies, n, t0);
  return (int)hipGetLastError();
}

// `complete` written by the device: word 0 of RAM A <- {ticks, BEGIN = 0}, then the library's sequence word (S/top_level.vhd:255-263)
NBL_HIDDEN int launch_mailbox_done_kernel(hipStream_t st, void* word0, unsigned* seq_word, const unsigned long long* t0, unsigned seq,
                                          unsigned clock_khz, unsigned rt_khz) {
  hipLaunchKernelGGL(mailbox_done_kernel, dim3(1), dim3(64), 0, st, (unsigned*)word0, seq_word, t0, seq, clock_khz, rt_khz);
  return (int)hipGetLastError();
}

NBL_HIDDEN int launch_rsqrt_selftest_kernel(unsigned first_bits, unsigned long long count, unsigned long long* out3) {
  const unsigned long long wgs = (count + 255) / 256;
  rsqrt_selftest_kernel<<<dim3((unsigned)(wgs < 16384 ? wgs : 16384)), dim3(256)>>>(first_bits, count, out3);
  return (int)hipGetLastError();
}

NBL_HIDDEN int launch_rsqrt_array_kernel(const float* x, float* y, int n, int ieee_only) {
  rsqrt_array_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(x, y, n, ieee_only ? 1 : 0);
  return (int)hipGetLastError();
}

}  // namespace nbl
