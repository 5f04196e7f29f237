/* nbody.h — C-ABI of the MI355X all-pairs N-body engine (libnbody_hip.so).
 *
 * Plain C99: pointers, ints, floats.  No HIP, torch or C++ types cross this
 * boundary.  The library behind it is hand-written HIP for gfx950
 * (mini_nbody_amd/csrc/: kernels.hip + context.cpp, comm.cpp, mailbox.cpp); there is no CPU fallback: every entry
 * point fails with NBODY_ERR_NO_DEVICE when no GPU is usable.
 *
 * WHAT EACH ENTRY POINT REPLACES.  The reference (/root/reference, a VHDL FPGA
 * design; "S/" = vec_add.srcs/sources_1/new/) has no function ABI: its only
 * boundary is a memory-mapped mailbox:
 *   - bodies in:  128-bit words {x, y, z, ignored} at word k = 1..N of RAM A    S/top_level.vhd:206-208, 238-240, 280
 *   - start:      word 0 = {bit 0 BEGIN, bits 46:32 NUM_PTS}                   S/top_level.vhd:184-185
 *   - forces out: 128-bit words {Fx, Fy, Fz, 0}: body k's at word k of RAM B,  S/compute_store.vhd:213, 221-242
 *                 word 0 never written
 *   - done:       word 0 rewritten with {ticks in bits 63:32}, BEGIN reads 0   S/top_level.vhd:146, 255-263
 *   - one request in flight; BEGIN ignored while busy                          S/top_level.vhd:180-186
 * The names bodyForce()/integrate() and the {pos, vel} layout come from
 * BASELINE.json's north_star (the host program they would belong to is not in
 * the reference tree); the 16-byte body word is the reference's.
 *
 * Conventions: every function returns 0 (NBODY_OK) on success; a positive value
 * < 1000 is a hipError_t, 1000..1999 an NBODY_ERR_*, 2000 + r an ncclResult_t r.
 * The caller owns host buffers, the library owns device buffers.  One context
 * per process, not thread-safe (the reference's single in-flight request); the one second thread the library knows — the mailbox's
 * service thread — locks every other caller out while it serves (nbody_mailbox_serve).
 */
#ifndef NBODY_H
#define NBODY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_OK 0
#define NBODY_ERR_NOT_INIT 1001
#define NBODY_ERR_ARG 1002
#define NBODY_ERR_NO_DEVICE 1003
#define NBODY_ERR_RCCL_LOAD 1004
#define NBODY_ERR_STATE 1005
#define NBODY_ERR_UNSUPPORTED 1006

/* Structure of arrays: two arrays of N 16-byte words.
 * pos[4i..4i+3] = {x, y, z, w}   (w is carried, never read by the force: S/top_level.vhd:206-208 ignores bits 127:96)
 * vel[4i..4i+3] = {vx, vy, vz, 0} */
typedef struct { float *pos, *vel; } BodySystem;
typedef struct { double *pos, *vel; } BodySystemD; /* fp64 configuration: N x 4 doubles (32-byte words) */

/* ---- options (nbody_set_option; all have working defaults) ---- */
enum {
  NBODY_OPT_VARIANT = 1,   /* how source bodies reach the lanes: NBODY_VARIANT_* */
  NBODY_OPT_IBLOCK = 2,    /* bodies per lane (register blocking) 1, 2, 4 or 8; 0 = auto */
  NBODY_OPT_JSUB = 3,      /* sub-segments per source slice; 0 = auto */
  NBODY_OPT_JSLICES = 4,   /* single-GPU only: number of source slices (to reproduce a P-GPU summation order bit for bit) */
  NBODY_OPT_ARITH = 5,     /* NBODY_ARITH_* */
  NBODY_OPT_SUM_ORDER = 6, /* NBODY_SUM_* */
  NBODY_OPT_TIMING = 7,    /* 1: HIP events around every force kernel (nbody_kernel_time) */
  NBODY_OPT_COMM = 8,      /* NBODY_COMM_* (multi-GPU) */
  NBODY_OPT_OVERLAP = 9,   /* multi-GPU: 1 = start on the rank's own slice while the others travel, then one launch over the
                              arrived slices (default); 2 = one launch per arriving slice, each released by that slice's
                              event; 0 = gather first, one launch */
  NBODY_OPT_SUM_BLOCK = 13, /* NBODY_SUM_BLOCKED: sources per level-1 block (multiple of 64; default 1024) */
  NBODY_OPT_XCD_MAP = 16,  /* 1: workgroups that share an XCD take the same source segments (launches with a multiple of 8 segment
                              rows), so each XCD's L2 fetches its segments once; 0: (blockIdx.x, blockIdx.y) = (row block, segment);
                              -1 (default): on for launches of >= 4096 row blocks (N = 1M: -39 % memory-side traffic at level time;
                              smaller launches measured slower with it).  Same results either way. */
  NBODY_OPT_ISA_LONG_BUFFERS = 15, /* NBODY_VARIANT_ISA: scalar buffers of 8 bodies instead of 4 (-1 = auto: when a launch has fewer
                              than 32 workgroups per CU; 0, 1 = force).  Same bits either way. */
  NBODY_OPT_FUSE_COMBINE = 14, /* 1: the segments' partial sums are added by the last wave to arrive, inside the force launch (one
                              launch per step); 0: a separate combine kernel; -1 (default): one launch when the launch has
                              >= 4 workgroups per CU, where the hand-off hides behind other workgroups, else two.  Same bits. */
  NBODY_OPT_GRAPH = 12,    /* nbody_step on one GPU replays a captured HIP graph of an even number of steps: 1 (default) = 32 steps per graph
                              when the call brings >= 64 (after the engine's first, eagerly launched step), 16 from 32, 8 from 16, else 2;
                              k >= 2 = k steps per graph (rounded down to even); 0: launch every kernel.  Same bits in every case. */
  NBODY_OPT_WAVES_PER_SIMD = 11, /* cap the force kernel's occupancy at k waves per SIMD (0 = no cap): tuning knob */
  NBODY_OPT_WSPLIT = 17,   /* 4 / 16: a workgroup owns 64 bodies and its 4 / 16 waves walk one piece of the source segment each; the
                              sums are added through LDS in ascending source order (a third level of the sum, like the reference's
                              16 partial sums + adder tree, S/fxyz.vhd:129-145, S/final_adder.vhd:88-104).  Same number of waves and
                              walk per wave from a quarter / a sixteenth of the global partial sums.  1: a workgroup owns
                              256 x IBLOCK bodies, every wave walks the whole segment (round 2's layout; the LDS / READLANE
                              kernels always).  -1 (default): where the kernel has it (SMEM and ISA deliveries, one body per lane),
                              16 when a rank owns <= 8192 bodies (fp32), else 4; 1 with NBODY_SUM_SEQ in fp32, which means ONE
                              sequential sum per segment.
                              With NBODY_SUM_FPGA16 (fp32) the split is of another kind and changes NO bit: -1 / 4 / 16 put the
                              reference's sixteen partial sums of a body on the sixteen waves of a workgroup (wave k walks the
                              sources k, k + 16, ... of the segment; wave 0 adds rotation and tree), 1 keeps all sixteen in one lane
                              (rounds 1-3).  Sixteen times the waves: the mailbox's maximum N = 32767 takes 0.51 ms per pass
                              instead of 2.56 (NBODY_INFO_WSPLIT then reports 16).  Round 6, still the same bits: the sixteen-wave form
                              stages its sources through LDS (NBODY_OPT_VARIANT = NBODY_VARIANT_SMEM keeps round 4's scalar delivery),
                              and a ONE-segment launch of fewer than 64 x CUs rows — the mailbox's faithful mode below N = 16384 — gives a
                              256-thread workgroup 16 rows x 16 partial sums, one (row, partial sum) per lane: four times the workgroups
                              where 64 rows per workgroup would leave CUs idle (N = 1024: 16 workgroups -> 64; NBODY_INFO_WSPLIT still
                              reports 16: sixteen partial sums per row, side by side). */
  NBODY_OPT_ISA_PHASE = 10 /* NBODY_VARIANT_ISA: which generated form of the hand-scheduled loop runs (tools/gen_force_loop.py).
                              1 = the product loop (default); 0 = the same instructions placed one 4-byte phase off (-27 %, kept
                              so that the placement effect can be re-measured).  fp64: 1 = the product loop (VALU instructions at
                              0 mod 8 bytes), 0 = one phase off (-1.2 %), 2 = eps and 3/8 from VGPR pairs (-0.9 %).
                              Every other value (fp32 2..18: experiment encodings of the same operations, and TIMING-ONLY forms
                              WITH WRONG RESULTS that price one part of the loop, profiles/r02_loop_diagnostics.md) exists only in
                              the diagnostic build `make diag` (libnbody_hip_diag.so, -DNBODY_DIAG_LOOPS); this library answers
                              NBODY_ERR_UNSUPPORTED. */
};
enum { NBODY_VARIANT_AUTO = 0,     /* ISA for the timed arithmetic (fp32, FMA3, blocked or sequential sum), else SMEM */
       NBODY_VARIANT_SMEM = 1,     /* wave-uniform scalar loads (s_load_dwordx16) into SGPRs: no LDS, no VALU cost */
       NBODY_VARIANT_LDS = 2,      /* `tile` bodies staged in LDS per workgroup, broadcast ds_read_b128 */
       NBODY_VARIANT_READLANE = 3, /* 64-body wave tile in VGPRs, v_readlane broadcast ("__shfl") */
       NBODY_VARIANT_ISA = 4       /* SMEM delivery with the inner loop hand-scheduled in gfx950 ISA (uniform 64-bit encodings,
                                      parity-safe registers); one body per lane, NBODY_ARITH_FMA3 only */ };
enum { NBODY_ARITH_FMA3 = 0,       /* d2 = fma(dx,dx,fma(dy,dy,fma(dz,dz,eps))): 11 VALU + v_rsq_f32 per pair.  THE TIMED MODE */
       NBODY_ARITH_REFERENCE = 1,  /* d2 = (dx*dx+dy*dy)+fma(dz,dz,eps): the RTL's rounding points, S/dxy.vhd:113-122, S/dzsoft.vhd:201-202, S/dxyz_soft.vhd:149-150 */
       NBODY_ARITH_STRICT = 2,     /* FMA3 with 1/sqrt rounded once from an fp64 evaluation instead of v_rsq_f32 (1 ulp):
                                      every operation is then IEEE-exact and the result is bit-identical to the CPU oracle.
                                      In an fp64 context: 1/sqrt as IEEE sqrt and divide instead of v_rsq_f64 + one third-order
                                      step — bit-identical to the oracle's fp64 evaluation in the configured summation order */
       NBODY_ARITH_REFERENCE_STRICT = 3 /* REFERENCE roundings + strict 1/sqrt (fp64 contexts have ONE d2 form — the fma-contracted one,
                                           as the oracle's fp64 evaluation — so there REFERENCE = FMA3 and REFERENCE_STRICT = STRICT) */ };
enum { NBODY_SUM_SEQ = 0,          /* one accumulator per segment, sources ascending (S/top_level.vhd:233-254) */
       NBODY_SUM_FPGA16 = 1,       /* 16 interleaved partials + pairwise tree (S/fxyz.vhd:129-184, S/final_adder.vhd:88-104) */
       NBODY_SUM_BLOCKED = 2       /* fp32 DEFAULT, THE TIMED MODE: two levels — blocks of NBODY_OPT_SUM_BLOCK consecutive sources
                                      are summed from zero, the block sums are added in ascending order.  The reference keeps
                                      16 partial sums + an adder tree for the same reason (S/fxyz.vhd:129-145,
                                      S/final_adder.vhd:88-104): one fp32 accumulator over 2^20 terms is off by 1e-4.
                                      fp64 contexts always sum sequentially. */ };
enum { NBODY_COMM_RING = 0,        /* P-1 ncclSend/ncclRecv ring steps, one event per arriving slice */
       NBODY_COMM_ALLGATHER = 1,   /* one in-place ncclAllGather (needs N divisible by the rank count, else ring) */
       NBODY_COMM_AUTO = 2,        /* default: ONE RCCL kernel per step, enqueued ahead of the force launch — ALLGATHER (RCCL's own ring over
                                      xGMI) when N is divisible by the rank count, DIRECT otherwise (profiles/r03_comm_under_load.md:
                                      a transfer kernel enqueued beside a force launch that fills every CU waits for wave slots, and
                                      P-1 dependent ring groups wait P-1 times) */
       NBODY_COMM_DIRECT = 3       /* one group of P-1 sends of the own slice and P-1 receives: one hop over all xGMI links */ };

/* ---- info keys (nbody_get_info) ---- */
enum { NBODY_INFO_N = 1, NBODY_INFO_N_LOCAL, NBODY_INFO_FIRST_BODY, NBODY_INFO_RANK, NBODY_INFO_NRANKS,
       NBODY_INFO_VARIANT, NBODY_INFO_IBLOCK, NBODY_INFO_JSUB, NBODY_INFO_NSEG, NBODY_INFO_DEVICE,
       NBODY_INFO_CU_COUNT, NBODY_INFO_CLOCK_KHZ, NBODY_INFO_FP64, NBODY_INFO_TILE, NBODY_INFO_STEPS_DONE,
       NBODY_INFO_SUM_ORDER, NBODY_INFO_SUM_BLOCK, NBODY_INFO_LAUNCHES_PER_STEP /* kernel launches one nbody_step() step takes */,
       NBODY_INFO_HAS_COMM /* 1: an RCCL communicator exists */,
       NBODY_INFO_WSPLIT /* resolved NBODY_OPT_WSPLIT: 1, 4 or 16 */, NBODY_INFO_ISA_PHASE, NBODY_INFO_LONG_BUFFERS /* option value, -1 = auto */,
       NBODY_INFO_XCD_MAP /* option value, -1 = auto */, NBODY_INFO_FUSE_COMBINE /* resolved: 1 = in-launch combine */,
       NBODY_INFO_COMM_FORM /* resolved NBODY_COMM_* of a multi-rank context, -1 with one rank */,
       NBODY_INFO_COMM_PRIORITY /* HIP priority of the transfer stream (0 = default priority) */,
       NBODY_INFO_DIAG_BUILD /* 1: this is libnbody_hip_diag.so (timing-only loop forms present) */,
       NBODY_INFO_MAILBOX_SERVED /* requests the mailbox's service thread has completed in this process */,
       NBODY_INFO_MAILBOX_SERVING /* 1: nbody_mailbox_serve(1, .) is in effect */ };

/* ---- lifetime ----
 * Replaces: power-up of the PL design + the ps_pl RAM allocation (S/top_level.vhd:100-117, 148-163). */

/* One process driving `ngpus` devices (ngpus = 1: device NBODY_DEVICE or 0).  tile: LDS tile in
 * bodies for NBODY_VARIANT_LDS (64..1024, multiple of 64; 0 = 256). */
int nbody_init(int n, int ngpus, int fp64, int tile);
/* One process per GPU (the torch.distributed / MPI launch): rank 0 calls nbody_unique_id() and
 * broadcasts the 128 bytes by its own means; every rank then calls nbody_init_rank().  Device =
 * NBODY_DEVICE, else LOCAL_RANK, else rank modulo the visible device count. */
int nbody_unique_id(void *uid128);
int nbody_init_rank(int n, int fp64, int tile, int rank, int nranks, const void *uid128);
/* Transport self-test on the communicator nbody_init_rank created (uid128 != NULL, any nranks >= 1): an in-place
 * all-gather of a patterned array in the configured NBODY_OPT_COMM form and one grouped ncclSend/ncclRecv ring step,
 * every received word checked.  Collective: every rank calls it.  *bytes_moved (may be NULL) = bytes this rank received. */
int nbody_comm_selftest(long long *bytes_moved);
/* The transfer plans of `vranks` VIRTUAL ranks (2..16; a job of that many ranks over this context's N bodies, ragged slices
 * included) in form NBODY_COMM_RING or NBODY_COMM_DIRECT, executed through real ncclSend/ncclRecv on the ONE-rank
 * communicator of nbody_init_rank(..., nranks = 1, uid): each virtual rank has its own array holding its own slice, every
 * receive is issued with the send its peer's plan pairs with it, and afterwards every array must hold all N words.  How a
 * one-GPU box runs the P > 1 offsets, byte counts and pairing of the data path. */
int nbody_comm_selftest_virtual(int vranks, int form, long long *bytes_moved);
/* NBODY_ARITH_STRICT in binary32 evaluates 1/sqrt to the value (float)(1.0 / sqrt((double)x)) — IEEE square root and divide in binary64,
 * rounded once — from eight binary32 operations, and falls back to that expression itself for the ~2^-16 of arguments whose value lies too
 * close to a rounding boundary to be decided that way (csrc/nbody_kernels.hpp rsqrt_strict_f32).  nbody_rsqrt_selftest() proves it on the
 * device: for each of the `count` binary32 BIT PATTERNS first_bits, first_bits + 1, ... (count <= 2^32 covers every float) it evaluates both
 * and counts the patterns where the eight-operation value was accepted and differs (*mismatches: must be 0; *first_bad = the smallest such
 * pattern) and the patterns sent to the IEEE form (*ieee_lanes).  Any pointer may be NULL.  Needs no context (it runs on the context's
 * first device, or on the device a one-GPU context would take).
 * nbody_strict_proof(): THE GATE.  That comparison over every positive normal binary32 (first_bits 0x00800000, count 0x7F000000: 10 ms),
 * on EVERY device of the open context (without a context: on the device a one-GPU context would take), cached per device for the
 * life of the process.  NBODY_OK: proved everywhere; NBODY_ERR_UNSUPPORTED: some device returned a different value (*mismatches,
 * *first_bad — either may be NULL — say how many and where).  The library calls it itself: nbody_set_option(NBODY_OPT_ARITH, a strict
 * mode) in an fp32 context and nbody_mailbox_open(., 1) answer NBODY_ERR_UNSUPPORTED and leave the arithmetic as it was unless every
 * device of the context has proved it — no host layer has to remember the check.
 * nbody_rsqrt_strict(): y[i] = that 1/sqrt of x[i], n values through host pointers, as the force kernels evaluate it (ieee_only = 0) or by
 * the IEEE expression alone (1) — the test surface that ties both to the CPU oracle. */
int nbody_rsqrt_selftest(unsigned first_bits, unsigned long long count, unsigned long long *mismatches, unsigned long long *ieee_lanes, unsigned *first_bad);
int nbody_rsqrt_strict(const float *x, float *y, int n, int ieee_only);
int nbody_strict_proof(unsigned long long *mismatches, unsigned *first_bad);
/* The transfer plan nbody_step() executes for rank `rank` of `nranks` over n bodies (form NBODY_COMM_RING or _DIRECT):
 * 7 values per send/receive pair {group, send_peer, send_first_word, send_words, recv_peer, recv_first_word, recv_words},
 * pairs of one group go into one ncclGroupStart/End.  Pure host arithmetic: needs no GPU and no context.  ops may be NULL
 * (count only); *n_ops = nranks - 1. */
int nbody_comm_plan(int form, int rank, int nranks, int n, long long *ops, int max_ops, int *n_ops);
/* One RCCL ring step (grouped ncclSend to rank+1 / ncclRecv from rank-1) of `bytes` on the transfer stream, timed with HIP
 * events: when = 0 alone, 1 enqueued just before a full force pass, 2 just after it (the force launch occupies every CU)
 * — enqueue to completion; 3 = the steady state of a multi-GPU step (ring step and the next force pass both released by
 * the end of the previous pass), 4 = the same with the hand-shake nbody_step() uses — from the previous pass's end to the
 * ring step's completion.  *force_ms = duration of the (last) force pass (0 for when = 0).  Needs a communicator. */
int nbody_comm_probe(long long bytes, int when, double *comm_ms, double *force_ms);
void nbody_shutdown(void);

int nbody_set_option(int key, int value);
int nbody_get_info(int key, long long *value);
const char *nbody_error_string(int code);

/* ---- state transfer (host <-> device mirrors) ----
 * Replaces: the PS writing bodies to RAM A words 1..N and reading RAM B (S/top_level.vhd:206-208; S/compute_store.vhd:242).
 * Arrays are always the FULL N bodies, on every rank; a rank keeps its own slice of vel. */
int nbody_upload(const BodySystem *host);
int nbody_download(BodySystem *host);
/* This rank's own bodies only: n_local words of pos and of vel (float or double words as the context was created),
 * no collective.  With one GPU it is the whole system. */
int nbody_download_slice(void *pos_words, void *vel_words);
int nbody_upload_d(const BodySystemD *host);
int nbody_download_d(BodySystemD *host);

/* ---- the path ---- */

/* v_i += dt * sum_j (r_j - r_i) * (|r_j - r_i|^2 + 1e-9f)^(-3/2), all j including i; pos is read-only.
 * Host pointers, n must equal the n of nbody_init.  (upload, force kernel + kick, download of vel.)
 * Replaces one full pass of the FSM, S/top_level.vhd:176-272, followed by the host's kick. */
int bodyForce(float *pos, float *vel, float dt, int n);
/* r_i += v_i * dt.  Host pointers. */
int integrate(float *pos, const float *vel, float dt, int n);
int bodyForce_d(double *pos, double *vel, double dt, int n);
int integrate_d(double *pos, const double *vel, double dt, int n);

/* Device-resident loop: nsteps x { bodyForce; integrate } on the uploaded state, no host round trip
 * per step; asynchronous (nbody_sync or nbody_download waits).  THE TIMED PATH. */
int nbody_step(float dt, int nsteps);
int nbody_step_d(double dt, int nsteps);
int nbody_sync(void);

/* Force-only entry point with the reference's word layouts: pos_words = N x {x, y, z, ignored}
 * (S/top_level.vhd:206-208), force_words = N x {Fx, Fy, Fz, 0} (S/compute_store.vhd:213, 242). */
int nbody_forces(const float *pos_words, float *force_words, int n);
int nbody_forces_d(const double *pos_words, double *force_words, int n);
/* Forces on `n_rows` bodies starting at `first_row`, from the state already on the device (row-sampled parity checks at
 * N = 1M).  first_row: in an nbody_init context (one process, one or several devices) the GLOBAL body index — the range may
 * span devices; in an nbody_init_rank context the row within this rank's own slice. */
int nbody_forces_rows(int first_row, int n_rows, float *force_words);
int nbody_forces_rows_d(int first_row, int n_rows, double *force_words);

/* ---- the reference's mailbox (its ONLY interface): the protocol the RTL intends ----
 * RAM A = capacity + 1 words of 16 bytes: word 0 = control {bit 0 BEGIN, bits 46:32 NUM_PTS}, words 1..N = {x, y, z, ignored}
 * (S/top_level.vhd:184-185, 206-208).  RAM B = capacity + 1 words: word k = {Fx, Fy, Fz, 0} of body k — the index the body has in RAM A —
 * and word 0 is NEVER written (S/compute_store.vhd:213, 221-242: write_we and STORE_PTR + 1 are registered on the same edge and write_addr
 * is formed from STORE_PTR combinationally, so the RAM samples we = 1 with the incremented address; the cycle model of those lines is
 * tests/test_fpga_store_model.py.  The signal's name, ZERO_PTR, :76-77, suggests the author meant word k - 1; the RTL does not do it.)
 * "The protocol the RTL intends", not "the RTL verbatim": the sequencer as written breaks that protocol in six places (NUM_PTS <= 3 is
 * never streamed, uninitialised pointers, `waiting` polls word 1, a stale BEGIN starts a second pass, NUM_PTS >= 32761 never ends,
 * `complete` overtakes the last block-group's stores) — shown cycle by cycle in tests/test_fpga_fsm_model.py, listed with what this
 * library does instead in INTEGRATION.md ("Departures from the RTL as written").  None of them is reproduced.
 *
 * nbody_mailbox_open(capacity, faithful): power-up of the PL block — ONE context that then serves any number of requests of any size.
 *   capacity = body words of the two RAMs, 1..32767 (= ram_depth - 1, S/top_level.vhd:45); 0 means 32767.  Replaces any open context.
 *   faithful = 1: the PL block's own bits — NBODY_ARITH_REFERENCE_STRICT (the RTL's rounding points, S/dxy.vhd:113-122, S/dzsoft.vhd:201-202,
 *   S/dxyz_soft.vhd:149-150, 1/sqrt rounded once) + NBODY_SUM_FPGA16 (sixteen partial sums, rotation, adder tree: S/fxyz.vhd:129-184,
 *   S/final_adder.vhd:88-104) + NBODY_OPT_JSUB 1 (one stream of all N sources per body, S/top_level.vhd:233-254) — granted only after the
 *   device has proved the strict 1/sqrt (nbody_strict_proof; NBODY_ERR_UNSUPPORTED and no context otherwise).  faithful = 0: the
 *   engine's timed arithmetic (every force within 1e-5 of the RTL's, not bit-equal).  nbody_shutdown() closes it.
 * nbody_mailbox_rams(): the context's own RAM images — pinned host memory the device reads (RAM A) and writes (RAM B, and word 0 of RAM A
 *   on completion) directly, the PS's view of the two block RAMs (S/top_level.vhd:100-117, 148-163).  A driver that fills *ram_a and passes
 *   these two pointers to nbody_mailbox_run moves no byte on the host; any other host buffers (of NUM_PTS + 1 words each) are accepted too
 *   and cost one host copy each way.  *capacity (may be NULL) = the largest NUM_PTS the context takes.  Works in any one-GPU fp32 context
 *   (nbody_init(n, 1, 0, .): capacity n).
 * nbody_mailbox_run(ram_a, ram_b, clock_khz): ONE request, synchronous (one request in flight, S/top_level.vhd:180-186).
 *   NUM_PTS is sampled from word 0 with every request (S/top_level.vhd:180-186): any value 0..capacity, request after request, on the
 *   same context — 1, 2 and 3 included (the RTL as written stores nothing for those: a stated departure).  Word 0 of RAM B and the words
 *   beyond NUM_PTS are never written (S/compute_store.vhd:221-242); NUM_PTS = 0 completes at once with RAM B untouched
 *   (S/top_level.vhd:189-192).  On return word 0 of ram_a is {ticks in bits 63:32, 0 elsewhere} — BEGIN reads 0.
 *   COMPLETION IS THE DEVICE'S, as in the PL block (S/top_level.vhd:121-146, 255-263): the last launch of a request rewrites word 0 of the
 *   pinned RAM A itself — ticks = 1 + the elapsed 1000-clock units of a `clock_khz` clock (0: 300 MHz) between the first wave that reads
 *   RAM A and that store, counted on the device's real-time counter — and clears BEGIN last, after every word of RAM B; the host only
 *   polls memory.  (NUM_PTS = 0, a refused request, a context over several ranks: the host writes word 0, ticks from its own clock.)
 *   Returns NBODY_ERR_STATE if BEGIN is not set (the FSM stays in `waiting`: nothing read, nothing written), NBODY_ERR_ARG if NUM_PTS
 *   exceeds the capacity (the RTL has no such case: its RAM always holds 32767 bodies).
 *   The context's N, options, uploaded state's size and step graph are as before on return (its position buffer is overwritten, as by
 *   nbody_forces).  A context over several devices or ranks keeps its fixed N (NUM_PTS must equal it).
 * nbody_mailbox_serve(on, clock_khz): the mailbox WITHOUT a call per request.  on = 1: a library thread takes the place of the PL block's
 *   FSM — it samples word 0 of the context's own RAM A (nbody_mailbox_rams) as the RTL does every clock (S/top_level.vhd:180-186) and runs
 *   each request it finds exactly as nbody_mailbox_run would; the device rewrites word 0 (ticks in bits 63:32, BEGIN cleared LAST, so
 *   whoever reads BEGIN = 0 also reads the ticks and RAM B).  The driver then only writes memory — bodies, then word 0 with NUM_PTS and
 *   BEGIN — and polls word 0, as the PS does.  A request the library cannot take (NUM_PTS beyond the capacity, a device error) completes
 *   with ticks = 0 and the error code in bits 127:96 of word 0, which the RTL always writes as 0.
 *   WHILE SERVING THE THREAD OWNS THE CONTEXT: every entry point that launches, copies or reconfigures (nbody_init*, nbody_mailbox_open,
 *   nbody_mailbox_run, nbody_set_option, nbody_upload*, nbody_download*, nbody_step*, nbody_sync, bodyForce*, integrate*, nbody_forces*,
 *   nbody_kernel_time, nbody_comm_*, nbody_device_ptr, the 1/sqrt self-tests) answers NBODY_ERR_STATE from any other thread;
 *   nbody_get_info reports the context's own N and configuration (never a request's), NBODY_INFO_MAILBOX_SERVED counts completed
 *   requests; nbody_mailbox_rams, nbody_error_string, nbody_mailbox_serve and nbody_shutdown stay available.  on = 0 (and
 *   nbody_shutdown) stops and joins the thread.  The idle thread backs off from spinning to 50-us naps after ~0.1 s without a request.
 *   One-GPU fp32 contexts only. */
int nbody_mailbox_open(int capacity, int faithful);
int nbody_mailbox_rams(void **ram_a, void **ram_b, int *capacity);
int nbody_mailbox_run(void *ram_a, void *ram_b, int clock_khz);
int nbody_mailbox_serve(int on, int clock_khz);

/* Multi-process transport without RCCL: call nbody_init_rank(..., uid128 = NULL), then register a function that
 * all-gathers a host array in place: on entry host_words[first..first+count) of this rank are valid (words of word_bytes
 * bytes, slices as nbody_get_info FIRST_BODY / N_LOCAL), on return all n_total words must be.  Returns 0 on success.
 * Slower (PCIe + the host framework's collective) but needs nothing from the GPU fabric; also what the two-process
 * GPU test uses on a one-GPU box. */
typedef int (*nbody_host_gather_fn)(void *user, void *host_words, int n_total, int word_bytes, int rank, int nranks);
int nbody_set_host_gather(nbody_host_gather_fn fn, void *user);

/* Sum of HIP-event durations of the force kernels since the last reset (NBODY_OPT_TIMING = 1). */
int nbody_kernel_time(double *ms_total, long long *launches, int reset);

/* Exposed communication: how long the compute stream sat waiting for arriving slices since the last reset (HIP events
 * around every such wait, NBODY_OPT_TIMING = 1).  0 when everything arrived while the own-slice kernel was running. */
int nbody_comm_time(double *wait_ms_total, long long *waits, int reset);

/* Device pointers of the resident state (for zero-copy interop with a framework that owns a view):
 * which = 0 current positions (full N), 1 velocities (own slice), 2 last forces (own slice). */
int nbody_device_ptr(int which, void **ptr, size_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_H */
