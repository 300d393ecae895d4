// tools/valu_issue2.hip -- companion of valu_issue.hip: issue cost and dependent latency of the multiply / convert / FP64 /
// transcendental / cross-lane instructions the descriptor and bundle-adjustment kernels lean on (gfx950).
// Same method: 256 CUs x W workgroups of 4 waves (W waves per SIMD), REPS x 64 inline-asm instructions per wave, one
// dependent chain or 8 independent chains; issue cost = kernel time x clock / (instructions x W).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_issue2.hip -o tools/bin/valu_issue2 ; run: valu_issue2 <sclk MHz>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum { MUL_LO = 0, MUL_U24, MUL_HI, MUL_F32, FMA_F32, CVT_I32_F32, CVT_F32_I32, RNDNE_F32, MUL_F64, ADD_F64, FMA_F64, RSQ_F64, RCP_F64,
       SQRT_F32, READLANE, DPP_MOV, BPERMUTE, LSHL_ADD_U64, PK_MUL_F32 };

template <int OP, bool DEP>
__global__ void __launch_bounds__(256) k_chain(int reps, unsigned* out, long long* cyc) {
  unsigned a[8];
  double d[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i + 3; d[i] = 1.0 + 1e-3 * (threadIdx.x + i); }
  const unsigned b = 0x00070003u + blockIdx.x;
  const double db = 1.0000001;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#define I32_2(ins) if (DEP) { REP64(asm volatile(ins " %0, %0, %1" : "+v"(a[0]) : "v"(b));) } else { REP8(asm volatile( \
      ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
#define I32_3(ins) if (DEP) { REP64(asm volatile(ins " %0, %0, %1, %1" : "+v"(a[0]) : "v"(b));) } else { REP8(asm volatile( \
      ins " %0, %0, %8, %8\n" ins " %1, %1, %8, %8\n" ins " %2, %2, %8, %8\n" ins " %3, %3, %8, %8\n" ins " %4, %4, %8, %8\n" ins " %5, %5, %8, %8\n" ins " %6, %6, %8, %8\n" ins " %7, %7, %8, %8" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));) }
#define I32_1(ins) if (DEP) { REP64(asm volatile(ins " %0, %0" : "+v"(a[0]));) } else { REP8(asm volatile( \
      ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7" \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));) }
#define F64_2(ins) if (DEP) { REP64(asm volatile(ins " %0, %0, %1" : "+v"(d[0]) : "v"(db));) } else { REP8(asm volatile( \
      ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8" \
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(db));) }
#define F64_3(ins) if (DEP) { REP64(asm volatile(ins " %0, %0, %1, %1" : "+v"(d[0]) : "v"(db));) } else { REP8(asm volatile( \
      ins " %0, %0, %8, %8\n" ins " %1, %1, %8, %8\n" ins " %2, %2, %8, %8\n" ins " %3, %3, %8, %8\n" ins " %4, %4, %8, %8\n" ins " %5, %5, %8, %8\n" ins " %6, %6, %8, %8\n" ins " %7, %7, %8, %8" \
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(db));) }
#define F64_1(ins) if (DEP) { REP64(asm volatile(ins " %0, %0" : "+v"(d[0]));) } else { REP8(asm volatile( \
      ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7" \
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));) }
    if (OP == MUL_LO) { I32_2("v_mul_lo_u32") }
    if (OP == MUL_U24) { I32_2("v_mul_u32_u24") }
    if (OP == MUL_HI) { I32_2("v_mul_hi_u32") }
    if (OP == MUL_F32) { I32_2("v_mul_f32") }
    if (OP == FMA_F32) { I32_3("v_fma_f32") }
    if (OP == CVT_I32_F32) { I32_1("v_cvt_i32_f32") }
    if (OP == CVT_F32_I32) { I32_1("v_cvt_f32_i32") }
    if (OP == RNDNE_F32) { I32_1("v_rndne_f32") }
    if (OP == SQRT_F32) { I32_1("v_sqrt_f32") }
    if (OP == MUL_F64) { F64_2("v_mul_f64") }
    if (OP == ADD_F64) { F64_2("v_add_f64") }
    if (OP == FMA_F64) { F64_3("v_fma_f64") }
    if (OP == RSQ_F64) { F64_1("v_rsq_f64") }
    if (OP == RCP_F64) { F64_1("v_rcp_f64") }
    if (OP == PK_MUL_F32) { F64_2("v_pk_mul_f32") }
    if (OP == LSHL_ADD_U64) { if (DEP) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d[0]) : "v"(db));) } else { REP8(asm volatile(
      "v_lshl_add_u64 %0, %0, 1, %8\nv_lshl_add_u64 %1, %1, 1, %8\nv_lshl_add_u64 %2, %2, 1, %8\nv_lshl_add_u64 %3, %3, 1, %8\n"
      "v_lshl_add_u64 %4, %4, 1, %8\nv_lshl_add_u64 %5, %5, 1, %8\nv_lshl_add_u64 %6, %6, 1, %8\nv_lshl_add_u64 %7, %7, 1, %8"
      : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(db));) } }
    if (OP == READLANE) {   // VGPR -> SGPR -> VGPR round trip (the broadcast idiom): v_readlane_b32 + v_mov_b32 from the SGPR
      unsigned s;
      if (DEP) { REP64(asm volatile("v_readlane_b32 %1, %0, 3\n s_nop 3\n v_add_u32 %0, %1, %0" : "+v"(a[0]), "=s"(s));) }
      else { REP8(asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %8, %1, 3\n v_readlane_b32 %8, %2, 3\n v_readlane_b32 %8, %3, 3\n"
                               "v_readlane_b32 %8, %4, 3\n v_readlane_b32 %8, %5, 3\n v_readlane_b32 %8, %6, 3\n v_readlane_b32 %8, %7, 3"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=s"(s));) }
    }
    if (OP == DPP_MOV) { if (DEP) { REP64(asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[0]));) } else { REP8(asm volatile(
      "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));) } }
    if (OP == BPERMUTE) {
      const unsigned addr = ((threadIdx.x ^ 1u) & 63u) << 2;
      if (DEP) { REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[0]) : "v"(addr));) }
      else { REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                               "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)"
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(addr));) }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
  for (int i = 0; i < 8; i++) acc += a[i] + (unsigned)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static unsigned* d_out;
static long long* d_cyc;
static double mhz = 2400.0;

template <int OP, bool DEP>
static void run(const char* name) {
  const int reps = 4000;
  for (int W : {1, 4, 8}) {
    const int grid = 256 * W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<OP, DEP>), dim3(grid), dim3(256), 0, 0, 10, d_out, d_cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_chain<OP, DEP>), dim3(grid), dim3(256), 0, 0, reps, d_out, d_cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid);
    (void)hipMemcpy(c.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : c) mean += (double)v;
    mean /= grid;
    const double instr = 64.0 * reps;
    printf("{\"op\": \"%s\", \"chain\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"wave_ticks_per_instr\": %.3f, "
           "\"issue_cycles_per_wave_instr\": %.3f}\n",
           name, DEP ? "dependent" : "8 independent", W, ms, mean / instr, ms * 1e-3 * mhz * 1e6 / (instr * W));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
}

int main(int argc, char** argv) {
  if (argc > 1) mhz = atof(argv[1]);
  (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned));
  (void)hipMalloc(&d_cyc, 256 * 8 * sizeof(long long));
#define R(id, name) run<id, true>(name); run<id, false>(name);
  R(MUL_LO, "v_mul_lo_u32") R(MUL_U24, "v_mul_u32_u24") R(MUL_HI, "v_mul_hi_u32") R(MUL_F32, "v_mul_f32") R(FMA_F32, "v_fma_f32")
  R(CVT_I32_F32, "v_cvt_i32_f32") R(CVT_F32_I32, "v_cvt_f32_i32") R(RNDNE_F32, "v_rndne_f32") R(SQRT_F32, "v_sqrt_f32")
  R(MUL_F64, "v_mul_f64") R(ADD_F64, "v_add_f64") R(FMA_F64, "v_fma_f64") R(RSQ_F64, "v_rsq_f64") R(RCP_F64, "v_rcp_f64")
  R(PK_MUL_F32, "v_pk_mul_f32") R(LSHL_ADD_U64, "v_lshl_add_u64")
  R(READLANE, "v_readlane_b32 (+ dependent v_add from the SGPR)") R(DPP_MOV, "v_add_u32_dpp row_shr:1") R(BPERMUTE, "ds_bpermute_b32")
  return 0;
}
