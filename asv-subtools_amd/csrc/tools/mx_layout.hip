// Developer probe (not part of libasv_amd.so): the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands, found by
// one-hot experiments instead of read off a table - which (lane, register, byte) of A / B holds which row / column and K index, and which
// lanes' scale bytes apply to which positions.
//
//   position = lane * 32 + reg * 4 + byte          (64 lanes x 8 registers x 4 bytes per operand)
//
// Experiments (one wave per configuration, all in one launch):
//   R: A one-hot at position p, B all ones            -> the row of D that is non-zero                       row_of_A[p]
//   C: B one-hot at position p, A all ones            -> the column                                          col_of_B[p]
//   K: A one-hot at p (rows 0 only), B all ones at (reg, byte) of ONE lane half -> which B (half, reg, byte) meets it      match
//   S: A one-hot at p, scale_a of lane s doubled      -> D doubles iff p lies in the block lane s scales     (same for B)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools_mx_layout tools/mx_layout.hip && ./tools_mx_layout
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Cfg {
  int a_mode, a_pos;        // 0 = all ones, 1 = one-hot at a_pos, 2 = ones at (reg, byte) = a_pos & 31 for lanes of half a_pos >> 5
  int b_mode, b_pos;
  int sa_lane, sb_lane;     // lane whose scale byte is 128 (x 2); -1: none
};

__device__ v8i build(int mode, int pos, int lane) {
  v8i v;
  for (int r = 0; r < 8; ++r) {
    uint32_t w = 0;
    for (int b = 0; b < 4; ++b) {
      bool one = mode == 0 || (mode == 1 && pos == lane * 32 + r * 4 + b) || (mode == 2 && (pos & 31) == r * 4 + b && (pos >> 5) == (lane >> 5));
      if (one) w |= 0x38u << (8 * b);                             // e4m3 1.0
    }
    v[r] = (int)w;
  }
  return v;
}

__global__ void run(const Cfg *cfg, float *out) {
  const Cfg c = cfg[blockIdx.x];
  const int lane = threadIdx.x;
  const v8i a = build(c.a_mode, c.a_pos, lane), b = build(c.b_mode, c.b_pos, lane);
  const int sa = lane == c.sa_lane ? 128 : 127, sb = lane == c.sb_lane ? 128 : 127;
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) {
    const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    out[(size_t)blockIdx.x * 1024 + row * 32 + col] = acc[r];
  }
}

int main() {
  std::vector<Cfg> cfgs;
  auto add = [&](Cfg c) { cfgs.push_back(c); return (int)cfgs.size() - 1; };
  const int R0 = (int)cfgs.size();
  for (int p = 0; p < 2048; ++p) add({1, p, 0, 0, -1, -1});
  const int C0 = (int)cfgs.size();
  for (int p = 0; p < 2048; ++p) add({0, 0, 1, p, -1, -1});
  // K: A one-hot at the 64 positions of lanes 0 and 32; B ones at (half, reg, byte)
  const int K0 = (int)cfgs.size();
  for (int la = 0; la < 2; ++la)
    for (int q = 0; q < 32; ++q)
      for (int hb = 0; hb < 64; ++hb) add({1, (la * 32) * 32 + q, 2, hb, -1, -1});
  // S: A one-hot at those 64 positions, scale_a doubled in lane 0 / 32 (B all ones); then the same for B against scale_b
  const int S0 = (int)cfgs.size();
  for (int la = 0; la < 2; ++la)
    for (int q = 0; q < 32; ++q)
      for (int s = 0; s < 2; ++s) add({1, (la * 32) * 32 + q, 0, 0, s * 32, -1});
  const int T0 = (int)cfgs.size();
  for (int lb = 0; lb < 2; ++lb)
    for (int q = 0; q < 32; ++q)
      for (int s = 0; s < 2; ++s) add({0, 0, 1, (lb * 32) * 32 + q, -1, s * 32});
  // which row does the scale of lane s apply to (A all ones, B all ones)
  const int U0 = (int)cfgs.size();
  for (int s = 0; s < 64; ++s) add({0, 0, 0, 0, s, -1});
  const int V0 = (int)cfgs.size();
  for (int s = 0; s < 64; ++s) add({0, 0, 0, 0, -1, s});

  Cfg *dc; float *dout;
  CK(hipMalloc(&dc, cfgs.size() * sizeof(Cfg))); CK(hipMalloc(&dout, cfgs.size() * 4096));
  CK(hipMemcpy(dc, cfgs.data(), cfgs.size() * sizeof(Cfg), hipMemcpyHostToDevice));
  run<<<(int)cfgs.size(), 64>>>(dc, dout);
  CK(hipDeviceSynchronize());
  std::vector<float> out(cfgs.size() * 1024);
  CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
  auto D = [&](int cfg, int row, int col) { return out[(size_t)cfg * 1024 + row * 32 + col]; };

  // R / C
  int bad = 0;
  for (int p = 0; p < 2048; ++p) {
    int row = -1, n = 0;
    for (int r = 0; r < 32; ++r) if (D(R0 + p, r, 0) != 0.f) { row = r; ++n; }
    if (n != 1 || row != ((p >> 5) & 31)) { if (bad < 8) printf("A position %d (lane %d): rows hit %d, row %d\n", p, p >> 5, n, row); ++bad; }
  }
  printf("A: row = lane & 31 for every (lane, reg, byte): %s\n", bad ? "NO" : "yes");
  bad = 0;
  for (int p = 0; p < 2048; ++p) {
    int col = -1, n = 0;
    for (int c = 0; c < 32; ++c) if (D(C0 + p, 0, c) != 0.f) { col = c; ++n; }
    if (n != 1 || col != ((p >> 5) & 31)) { if (bad < 8) printf("B position %d (lane %d): columns hit %d, column %d\n", p, p >> 5, n, col); ++bad; }
  }
  printf("B: column = lane & 31 for every (lane, reg, byte): %s\n", bad ? "NO" : "yes");
  // K: the B (half, reg, byte) that meets A (half la, reg, byte)
  printf("K pairing (A half.reg.byte -> B half.reg.byte):\n");
  int same = 0;
  for (int la = 0; la < 2; ++la)
    for (int q = 0; q < 32; ++q) {
      int hit = -1, n = 0;
      for (int hb = 0; hb < 64; ++hb) if (D(K0 + (la * 32 + q) * 64 + hb, 0, 0) != 0.f) { hit = hb; ++n; }
      if (n == 1 && hit == la * 32 + q) ++same;
      else printf("  A %d.%d.%d -> %d hit(s), B %d.%d.%d\n", la, q >> 2, q & 3, n, hit >> 5, (hit & 31) >> 2, hit & 3);
    }
  printf("  %d of 64 positions pair with the SAME (half, reg, byte) of B\n", same);
  // S
  printf("scale_a: A position (half.reg.byte) doubled by lane 0 / lane 32:\n");
  for (int la = 0; la < 2; ++la) {
    printf("  half %d: ", la);
    for (int q = 0; q < 32; ++q) {
      const float v0 = D(S0 + ((la * 32 + q) * 2 + 0), 0, 0), v1 = D(S0 + ((la * 32 + q) * 2 + 1), 0, 0);
      printf("%c", v0 == 2.f && v1 == 1.f ? '0' : (v0 == 1.f && v1 == 2.f ? '3' : '?'));
    }
    printf("   (per byte position reg*4+byte; '0' = lane 0's scale, '3' = lane 32's)\n");
  }
  printf("scale_b: B position doubled by lane 0 / lane 32:\n");
  for (int lb = 0; lb < 2; ++lb) {
    printf("  half %d: ", lb);
    for (int q = 0; q < 32; ++q) {
      const float v0 = D(T0 + ((lb * 32 + q) * 2 + 0), 0, 0), v1 = D(T0 + ((lb * 32 + q) * 2 + 1), 0, 0);
      printf("%c", v0 == 2.f && v1 == 1.f ? '0' : (v0 == 1.f && v1 == 2.f ? '3' : '?'));
    }
    printf("\n");
  }
  bad = 0;
  for (int s = 0; s < 64; ++s) {
    for (int r = 0; r < 32; ++r) {
      const float want = (r == (s & 31)) ? 96.f : 64.f;
      if (D(U0 + s, r, 5) != want) { if (bad < 6) printf("scale_a lane %d: row %d = %g\n", s, r, D(U0 + s, r, 5)); ++bad; }
    }
    for (int c = 0; c < 32; ++c) {
      const float want = (c == (s & 31)) ? 96.f : 64.f;
      if (D(V0 + s, 5, c) != want) { if (bad < 12) printf("scale_b lane %d: column %d = %g\n", s, c, D(V0 + s, 5, c)); ++bad; }
    }
  }
  printf("scale of lane s applies to row / column s & 31, one 32-deep half of K: %s\n", bad ? "NO" : "yes");
  return 0;
}
