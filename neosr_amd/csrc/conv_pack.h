// conv_pack.h — packed weight images for the direct-to-LDS 3x3 kernel (internal to libneosr_amd).
#pragma once
#include <cstdint>

namespace neosr_pack {

constexpr int IMG_FLOATS = 9 * 4 * 32 * 4;  // one (32-n block, 16-k chunk) slab: 4608 floats = 18 KB
constexpr int MAX_SEG = 5;

// One source convolution contributing the reduction rows [k_lo, k_lo + k_cnt) of an image.
struct Seg {
  const float* w;  // canonical (w_cout, w_cin, 3, 3)
  int32_t w_cin;
  int32_t k_lo, k_cnt;
  int32_t n_lo;  // image column n reads source row/column n_lo + n
};

// dst[nblk][chunk][tap][kq][n32][4]; mode FWD: element = w[n_lo + n, k - k_lo, tap];
// mode DGRAD: element = w[k - k_lo, n_lo + n, 8 - tap].
struct Image {
  float* dst;
  int32_t N, K, mode, nseg;
  Seg seg[MAX_SEG];
};

inline int64_t image_floats(int N, int K) {
  return (int64_t)((N + 31) / 32) * ((K + 15) / 16) * IMG_FLOATS;
}

// Winograd F(2x2,3x3) image (conv_wino.hip): [nblk][chunk 16 k][pos 16][k quad 4][n 32][4] floats, element =
// (G g G^T)[pos] of the same (n, k) the direct image holds taps of; 32 KB per (n-block, chunk)
constexpr int WINO_IMG_FLOATS = 16 * 4 * 32 * 4;
inline int64_t wino_image_floats(int N, int K) {
  return (int64_t)((N + 31) / 32) * ((K + 15) / 16) * WINO_IMG_FLOATS;
}

// Winograd F(4x4,3x3) image (conv_wino4.hip): [nblk][chunk 32 k][pos 36][k parity 2][cout block 2][k quad 4][cout 16][4]
// floats, element = (G g G^T)[pos] (6x6, computed in float64); 144 KB per (n-block, chunk)
constexpr int WINO4_IMG_FLOATS = 36 * 2 * 2 * 256;
inline int64_t wino4_image_floats(int N, int K) {
  return (int64_t)((N + 31) / 32) * ((K + 31) / 32) * WINO4_IMG_FLOATS;
}

constexpr int BATCH = 24;  // images per launch (passed by value as kernel arguments: 3.4 KB)
struct Batch {
  Image im[BATCH];
};

// `images` is a HOST array; ceil(n / BATCH) launches, nothing is copied to the device.
int launch(const Image* images, int n, void* stream);
int launch_wino(const Image* images, int n, void* stream);  // same descriptors, Winograd images
int launch_wino4(const Image* images, int n, void* stream); // same descriptors, Winograd F(4x4,3x3) images

}  // namespace neosr_pack
