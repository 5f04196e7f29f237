// nbody_args.hpp — what the host side and the kernels agree on: the launch argument block (ForceArgs), the workgroup shapes and the
// cuts of the source set into slices, segments and pieces.  Included by nbody_kernels.hpp (device) and nbody_internal.hpp (host);
// part of the kernel source: bench.py's kernel_source_sha() hashes it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbk {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;            // 4 waves, one per SIMD
// threads of a workgroup whose WS waves share 64 rows and split the segment (ForceArgs::wsplit): 256 (WS = 1 or 4), 1024 (16)
constexpr int wg_threads(int WS) { return WS > 1 ? 64 * WS : kBlock; }
constexpr uint32_t kSoftBits = 0x3089705Fu;  // S/dzsoft.vhd:177

// Address space 4 = constant: a load through it with a wave-uniform address is
// always selected as s_load_* (scalar cache), never as a vector load.
#define NB_CONST __attribute__((address_space(4)))

constexpr int kFinishStore = 0;    // store the segment's partial sum; combine_kernel follows
constexpr int kFinishDirect = 1;   // one segment: the kernel's own sum is the force
constexpr int kFinishLast = 2;     // store the partial sum (write-through); the last workgroup to arrive combines

struct ForceArgs {
  const void* src;      // all N source bodies (16-B or 32-B words), ascending
  const void* rows;     // the rank's own bodies: rows[i] = src[first_body + i]
  void* partial;        // [nseg][part_stride] words {Fx,Fy,Fz,0}, rows counted from row0 (launch-relative)
  void* vel;            // [n_rows]
  void* pos_next_rows;  // [n_rows]
  void* force_out;      // [n_rows] {Fx,Fy,Fz,0} (S/compute_store.vhd:242) or null
  unsigned* tickets;    // one arrival counter per 64 rows counted from row0 (kFinishLast); zero between steps
  int n_src;            // N
  int n_rows;           // bodies owned by this rank
  int row0;             // first row handled by this launch (nbody_forces_rows)
  int row_count;        // rows handled by this launch
  int nslices, sub;     // segmentation of the sources: nslices slices (one per rank), `sub` pieces each
  int slice_start;      // blockIdx.y / sub = 0 maps to this slice; then descending modulo nslices (ring arrival order)
  int nseg;             // nslices * sub: partial sums per row once every launch of the step has run
  int finish;           // kFinish*
  int do_kick, do_drift;  // what to do with the finished force: v += dt*F, then r' = r + v*dt
  int sum_block;        // K sources per level-1 block; 0 = one sequential sum per segment
  int long_buffers;     // ISA variant: 8-body scalar buffers (launches with < 4 waves per SIMD, see tools/gen_force_loop.py)
  int xcd_map;          // 1: workgroups that share an XCD (linear id mod 8) take the same source segments, see block_segment()
  int fpga16;           // 1: S/fxyz.vhd:129-184 + S/final_adder.vhd:88-104 summation order inside a segment
  int wsplit;           // 1: a workgroup owns 256*R rows and its four waves walk the same segment for different rows;
                        // 4: a workgroup owns 64 rows and wave w walks piece w of 4 of the segment for those same rows — the four
                        //    sums are added through LDS in ascending source order (a third level of the sum), so a launch of
                        //    the same workgroup count has a quarter of the global partial sums, tickets and last-arriver rounds;
                        // 16: the same with workgroups of 16 waves (1024 threads): with one segment per slice a workgroup walks
                        //    ALL sources of its 64 rows and finishes them itself — no global partial sums at all (mid N, where
                        //    the sources fit an XCD's L2)
  int part_stride;      // words between two segments' rows in `partial`: row_count rounded up to 64, so that the 1 KiB (2 KiB in
                        // fp64) regions of different waves/workgroups never share a cache line
  float dt;
  double dt64;
  // APPENDED (the hand-scheduled kernels read the fields above at fixed offsets): where the launch's first wave stamps the real-time counter
  // — the start of a mailbox request's tick count when the request has no ingest launch (force_fpga16r_f32 reading RAM A itself); null otherwise
  unsigned long long* t0_stamp;
};

// slice q of P over n: [first(q), first(q+1)), balanced
__device__ __host__ inline int slice_first(int q, int n, int P) {
  int base = n / P, rem = n % P;
  return q * base + (q < rem ? q : rem);
}
// segment (q, t): piece t of `sub` of slice q
__device__ __host__ inline void segment_bounds(int q, int t, int n, int P, int sub, int* jb, int* je) {
  int f0 = slice_first(q, n, P), f1 = slice_first(q + 1, n, P);
  int len = f1 - f0;
  int piece = (len + sub - 1) / sub;
  int b = f0 + t * piece;
  int e = b + piece;
  if (b > f1) b = f1;
  if (e > f1) e = f1;
  *jb = b; *je = e;
}

// piece w of `ws` of the segment [jb, je): what wave w of a workgroup walks when ForceArgs::wsplit = ws (same cut as
// a slice into pieces: ceil(len / ws) sources each, the last ones possibly shorter or empty)
__device__ __host__ inline void piece_bounds(int jb, int je, int w, int ws, int* pb, int* pe) {
  int piece = (je - jb + ws - 1) / ws;
  int b = jb + w * piece;
  int e = b + piece;
  if (b > je) b = je;
  if (e > je) e = je;
  *pb = b; *pe = e;
}

}  // namespace nbk
