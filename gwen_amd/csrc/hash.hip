// Content checksum of a device buffer -- the key of the prepared-graph cache.
//
// The reference's loops hand a NEW edge_index tensor to every forward (NeighborLoader batches:
// /root/reference/src/gwen/models_gnn.py:351-360, :434-443) although on its complete member graph every
// batch carries the same edges; torch-geometric re-normalises per layer regardless.  Here the prepared
// graph (K1) is cached, and a tensor the cache has not seen by identity is looked up by CONTENT: two
// independent 64-bit position-dependent sums over its 8-byte words (128 bits), one small launch pair, 16
// bytes read back.  Deterministic (fixed-order two-stage reduction, no atomics).
#include "common.h"

namespace {

constexpr int kHashBlocks = 512;

__device__ inline uint64_t mix64(uint64_t z) {       // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_hash_partial(const uint64_t *__restrict__ w, int64_t n_words,
                                                      const uint8_t *__restrict__ tail, int tail_bytes,
                                                      uint64_t *__restrict__ partial) {
  __shared__ uint64_t sa[256], sb[256];
  uint64_t a = 0, b = 0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * 256) {
    const uint64_t v = w[i];
    a += mix64(v + 0x9e3779b97f4a7c15ULL * (uint64_t)(i + 1));
    b += mix64((v ^ 0xd6e8feb86659fd93ULL) + 0xc2b2ae3d27d4eb4fULL * (uint64_t)(i + 1));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && tail_bytes > 0) {
    uint64_t v = 0;
    for (int k = 0; k < tail_bytes; ++k) v |= (uint64_t)tail[k] << (8 * k);
    a += mix64(v + 0x9e3779b97f4a7c15ULL * (uint64_t)(n_words + 1));
    b += mix64((v ^ 0xd6e8feb86659fd93ULL) + 0xc2b2ae3d27d4eb4fULL * (uint64_t)(n_words + 1));
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sa[0]; partial[2 * blockIdx.x + 1] = sb[0]; }
}

__global__ __launch_bounds__(256) void k_hash_final(const uint64_t *__restrict__ partial, int n_blocks,
                                                    uint64_t bytes, uint64_t *__restrict__ out) {
  __shared__ uint64_t sa[256], sb[256];
  uint64_t a = 0, b = 0;
  for (int i = threadIdx.x; i < n_blocks; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = mix64(sa[0] ^ bytes); out[1] = mix64(sb[0] + bytes); }
}

}  // namespace

extern "C" int64_t gwen_checksum_workspace_bytes(void) { return (int64_t)kHashBlocks * 16; }

extern "C" int gwen_checksum128(const void *data, int64_t bytes, uint64_t *out, void *workspace,
                                int64_t workspace_bytes, gwen_stream_t stream_) {
  if (bytes < 0 || !out || !workspace || workspace_bytes < gwen_checksum_workspace_bytes())
    return GWEN_EINVAL;
  if (bytes > 0 && (!data || !gwen_aligned(data, 8))) return GWEN_EINVAL;
  hipStream_t st = gwen_stream(stream_);
  const int64_t n_words = bytes / 8;
  const int tail = (int)(bytes - 8 * n_words);
  int64_t blocks = (n_words + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > kHashBlocks) blocks = kHashBlocks;
  uint64_t *partial = static_cast<uint64_t *>(workspace);
  k_hash_partial<<<(unsigned)blocks, 256, 0, st>>>(static_cast<const uint64_t *>(data), n_words,
                                                   static_cast<const uint8_t *>(data) + 8 * n_words, tail,
                                                   partial);
  GWEN_LAUNCH_CHECK();
  k_hash_final<<<1, 256, 0, st>>>(partial, (int)blocks, (uint64_t)bytes, out);
  GWEN_LAUNCH_CHECK();
  return GWEN_OK;
}
