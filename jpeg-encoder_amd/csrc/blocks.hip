// Per-block kernels behind the component-level entry points of the C-ABI: the
// granularity at which the reference's components work (one 8x8 block-channel
// per call: ColourMapper::map, DiscreteCosineTransformer::transform,
// Quantiser::quantise, EntropyEncoder::encode/decode).  They exist so that the
// host classes can offer those methods with all arithmetic on the GPU, and as a
// test surface; they run the same device functions as the fused image kernels.
// One thread per block -- these are not throughput kernels.
#include <hip/hip_runtime.h>

#include "dct_device.h"
#include "huffman_device.h"
#include "kernels.h"
#include "tables.h"

namespace jea {

namespace {

// DiscreteCosineTransformer::transform: level shift (discrete_cosine_transform.cpp:13-19)
// + SeparatedDiscreteCosineTransformer::applyTransform (:77-88).  out[8v+u].
__global__ void dct_forward_blocks_kernel(const uint8_t* __restrict__ in, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = in + i * 64;
    float row[8][8];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        float f[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) f[x] = (float)p[8 * y + x] - 128.0f;
        fdct_row(f, y, row);
    }
    float* o = out + i * 64;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float col[8];
        fdct_col(row, u, col);
#pragma unroll
        for (int v = 0; v < 8; ++v) o[8 * v + u] = col[v];
    }
}

// DiscreteCosineTransformer::inverseTransform (:8-11, :90-101, :21-27).
__global__ void dct_inverse_blocks_kernel(const float* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float deq[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) deq[k] = in[i * 64 + k];
    uint8_t px[64];
    idct_block(deq, px);
#pragma unroll
    for (int k = 0; k < 64; ++k) out[i * 64 + k] = px[k];
}

// Quantiser::quantise (quantiser.cpp:57-70), natural order in and out; the same
// division-free quotient as the fused kernel.
__global__ void quantise_blocks_kernel(const float* __restrict__ in, size_t n, QuantTables q, int use_lum,
                                       int16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* qt = use_lum ? q.lum_t : q.chr_t;
    const float* rqt = use_lum ? q.rlum_t : q.rchr_t;
    int qv[64];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float o[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) o[v] = in[i * 64 + 8 * v + u];
        float2 c[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) c[v] = make_float2(rqt[8 * u + v], qt[8 * u + v]);   // tables are stored [u][v]
        quantise_col(o, c, u, qv);
    }
#pragma unroll
    for (int k = 0; k < 64; ++k) out[i * 64 + k] = (int16_t)qv[k];
}

// Quantiser::dequantise (quantiser.cpp:72-85).
__global__ void dequantise_blocks_kernel(const int16_t* __restrict__ in, size_t n, QuantTables q, int use_lum,
                                         float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 64) return;
    const float* qt = use_lum ? q.lum : q.chr;
    out[i] = (float)in[i] * qt[i & 63];
}

// EntropyEncoder::encode for independent block-channels (entropy_encoder.cpp:3-7,
// 94-120, 696-765): coefficients already in zig-zag order, explicit previous DC
// and table class per block.  bits[i][0..51] receives the MSB-first string.
__global__ void huffman_encode_blocks_kernel(const int16_t* __restrict__ zz, const int16_t* __restrict__ prev_dc,
                                             const uint8_t* __restrict__ is_lum, size_t n,
                                             const HuffTables* __restrict__ huff, uint16_t* __restrict__ lengths,
                                             uint32_t* __restrict__ bits, int32_t* __restrict__ status) {
    __shared__ HuffLds h;
    load_huff<64>(h, huff);   // launched with 64 threads
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BlockRegs blk;
    load_block(blk, zz + i * 64);
    const int cls = is_lum[i] ? 0 : 1;
    SlotSink sink;
    sink.slot = bits + i * kSlotWords;
    sink.stride = 1;
    const bool ok = encode_block(blk, (int)prev_dc[i], h.dc[cls], h.ac[cls], sink);
    sink.finish();
    if (!ok) atomicExch(status, JEA_ERR_RANGE);
    lengths[i] = (uint16_t)sink.bits;
}

// EntropyEncoder::decode for one block-channel (entropy_encoder.cpp:9-13,
// 703-712, 767-866, 122-144) from an already unstuffed stream, starting at an
// arbitrary bit.  Serial by nature: one lane works, the wave stages the bytes.
__global__ __launch_bounds__(64) void huffman_decode_block_kernel(const uint8_t* __restrict__ stream, uint64_t len,
                                                                  uint64_t start_bit, int is_lum, int last_dc_in,
                                                                  const DecodeTables* __restrict__ tables,
                                                                  int16_t* __restrict__ zz_out,
                                                                  uint64_t* __restrict__ end_bit,
                                                                  int32_t* __restrict__ last_dc_out,
                                                                  int32_t* __restrict__ status) {
    __shared__ DecodeTables t;
    __shared__ __align__(16) uint8_t win[kStreamWin + 16];
    __shared__ int16_t zz[64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < sizeof(DecodeTables) / 4; i += 64)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tables)[i];
    const uint64_t first_byte = start_bit >> 3;
    for (uint32_t i = lane; i < kStreamWin + 16; i += 64) win[i] = (first_byte + i < len) ? stream[first_byte + i] : 0;
    zz[lane] = 0;
    __syncthreads();
    if (lane == 0) {
        // scan_end = 0: stuffing has been removed by the caller (encoder.cpp:49), nothing is skipped here
        SerialReader rd{win, first_byte, first_byte, 0, len, 0, 0, 0};
        rd.refill();
        rd.skip((int)(start_bit & 7));
        int last = last_dc_in;
        const int cls = is_lum ? 0 : 1;
        const int err = decode_block(rd, t.tab[2 * cls], t.tab[2 * cls + 1], last, zz);
        if (err) atomicExch(status, err);
        // bits consumed = bytes taken into the buffer * 8 - bits still buffered
        *end_bit = rd.pos * 8 - (uint64_t)rd.nbits;
        *last_dc_out = last;
    }
    __syncthreads();
    zz_out[lane] = zz[lane];
}

unsigned blocks_for(size_t n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

void launch_dct_forward_blocks(hipStream_t s, const uint8_t* d_in, size_t n, float* d_out) {
    dct_forward_blocks_kernel<<<blocks_for(n, 64), 64, 0, s>>>(d_in, n, d_out);
}
void launch_dct_inverse_blocks(hipStream_t s, const float* d_in, size_t n, uint8_t* d_out) {
    dct_inverse_blocks_kernel<<<blocks_for(n, 64), 64, 0, s>>>(d_in, n, d_out);
}
void launch_quantise_blocks(hipStream_t s, const float* d_in, size_t n, const QuantTables& q, bool use_lum,
                            int16_t* d_out) {
    quantise_blocks_kernel<<<blocks_for(n, 64), 64, 0, s>>>(d_in, n, q, use_lum ? 1 : 0, d_out);
}
void launch_dequantise_blocks(hipStream_t s, const int16_t* d_in, size_t n, const QuantTables& q, bool use_lum,
                              float* d_out) {
    dequantise_blocks_kernel<<<blocks_for(n * 64, 256), 256, 0, s>>>(d_in, n, q, use_lum ? 1 : 0, d_out);
}
void launch_huffman_encode_blocks(hipStream_t s, const int16_t* d_zz, const int16_t* d_prev_dc, const uint8_t* d_is_lum,
                                  size_t n, const HuffTables* d_huff, uint16_t* d_lengths, uint32_t* d_bits,
                                  int32_t* d_status) {
    huffman_encode_blocks_kernel<<<blocks_for(n, 64), 64, 0, s>>>(d_zz, d_prev_dc, d_is_lum, n, d_huff, d_lengths, d_bits,
                                                                  d_status);
}
void launch_huffman_decode_block(hipStream_t s, const uint8_t* d_stream, uint64_t len, uint64_t start_bit, bool is_lum,
                                 int last_dc, const DecodeTables* d_tables, int16_t* d_zz, uint64_t* d_end_bit,
                                 int32_t* d_last_dc, int32_t* d_status) {
    huffman_decode_block_kernel<<<1, 64, 0, s>>>(d_stream, len, start_bit, is_lum ? 1 : 0, last_dc, d_tables, d_zz,
                                                 d_end_bit, d_last_dc, d_status);
}

}  // namespace jea
