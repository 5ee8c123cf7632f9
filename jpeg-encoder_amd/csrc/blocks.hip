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
        SerialReader rd{win, first_byte, first_byte, len, 0, 0, 0};
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

// HuffmanEncoder::applyFinalEncoding (entropy_encoder.cpp:696-701, 715-765) for one block-channel whose
// run-length form the caller supplies: the DC difference, then every (run, value) entry exactly as given --
// (0, 0) is the end-of-block code, (15, 0) the zero-run code, any other entry with value 0 is the
// reference's "Invalid runtime encoding encountered." (:748).  The run is narrowed to 8 bits first, as the
// reference does (:733: a run of 256 with value 0 is an end of block).  Entries the tables have no code for (a
// narrowed run above 15 with a value, an amplitude of more than 10 bits, a DC difference of more than 11) index
// past the reference's tables, which is undefined there; here they are reported as the same error.  bits[] receives the MSB-first string, zero padded.
__global__ __launch_bounds__(64) void huffman_encode_rle_kernel(int dc_difference, const uint16_t* __restrict__ runs,
                                                                const int16_t* __restrict__ values, uint32_t n,
                                                                int is_lum, const HuffTables* __restrict__ huff,
                                                                uint32_t* __restrict__ bits, uint32_t* __restrict__ length,
                                                                int32_t* __restrict__ status) {
    __shared__ HuffLds h;
    load_huff<64>(h, huff);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int cls = is_lum ? 0 : 1;
    const uint32_t* dc_tab = h.dc[cls];
    const uint32_t* ac_tab = h.ac[cls];
    SlotSink sink;
    sink.slot = bits;
    sink.stride = 1;
    bool ok = true;
    {
        const uint32_t amp = (uint32_t)(dc_difference < 0 ? -dc_difference : dc_difference) & 0xffffu;
        uint32_t s = bit_width(amp);
        if (s > 11) { ok = false; s = 11; }
        const uint32_t e = dc_tab[s];
        const uint32_t extra = (dc_difference > 0 ? amp : ~amp) & ((1u << s) - 1u);
        sink.put(((e & 0xffffu) << s) | extra, (e >> 16) + s);
    }
    for (uint32_t i = 0; i < n && ok; ++i) {
        const uint32_t run = runs[i] & 0xffu;   // uint8_t runLengthRRRR = acCoeff.m_runLength (:733)
        const int v = values[i];
        const uint32_t amp = (uint32_t)(v < 0 ? -v : v) & 0xffffu;
        const uint32_t s = bit_width(amp);
        if (s == 0) {
            if (run != 0 && run != 15) { ok = false; break; }
            const uint32_t e = ac_tab[run == 0 ? 0x00 : 0xF0];
            sink.put(e & 0xffffu, e >> 16);
        } else {
            if (run > 15 || s > 10) { ok = false; break; }
            const uint32_t e = ac_tab[(run << 4) | s];
            const uint32_t extra = (v > 0 ? amp : ~amp) & ((1u << s) - 1u);
            sink.put(((e & 0xffffu) << s) | extra, (e >> 16) + s);
        }
    }
    sink.finish();
    *length = sink.bits;
    if (!ok) atomicExch(status, JEA_ERR_RANGE);
}

// HuffmanEncoder::removeFinalEncoding (entropy_encoder.cpp:703-712, 767-866) for one block-channel of an
// unstuffed stream: the DC difference and the (run, value) entries up to and including the end-of-block
// entry, or until 63 coefficient positions are covered.
__global__ __launch_bounds__(64) void huffman_decode_rle_kernel(const uint8_t* __restrict__ stream, uint64_t len,
                                                                uint64_t start_bit, int is_lum,
                                                                const DecodeTables* __restrict__ tables,
                                                                int32_t* __restrict__ dc_difference,
                                                                uint16_t* __restrict__ runs, int16_t* __restrict__ values,
                                                                uint32_t* __restrict__ n_entries,
                                                                uint64_t* __restrict__ end_bit, int32_t* __restrict__ status) {
    __shared__ DecodeTables t;
    __shared__ __align__(16) uint8_t win[kStreamWin + 16];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < sizeof(DecodeTables) / 4; i += 64)
        reinterpret_cast<uint32_t*>(&t)[i] = reinterpret_cast<const uint32_t*>(tables)[i];
    const uint64_t first_byte = start_bit >> 3;
    for (uint32_t i = lane; i < kStreamWin + 16; i += 64) win[i] = (first_byte + i < len) ? stream[first_byte + i] : 0;
    __syncthreads();
    if (lane != 0) return;
    SerialReader rd{win, first_byte, first_byte, len, 0, 0, 0};
    rd.refill();
    rd.skip((int)(start_bit & 7));
    const int cls = is_lum ? 0 : 1;
    const DecodeTable& dct = t.tab[2 * cls];
    const DecodeTable& act = t.tab[2 * cls + 1];
    int err = 0;
    rd.refill();
    const int s = decode_symbol(rd, dct);
    int diff = 0;
    if (s < 0 || s > 11) err = JEA_ERR_BAD_STREAM;
    else if (s > 0) { rd.refill(); diff = read_amplitude(rd, s); }
    uint32_t n = 0;
    int processed = 0;
    while (!err) {
        rd.refill();
        const int sym = decode_symbol(rd, act);
        if (sym < 0) { err = JEA_ERR_BAD_STREAM; break; }
        int run, val;
        if (sym == 0x00) { run = 0; val = 0; }
        else if (sym == 0xF0) { run = 15; val = 0; }
        else {
            run = sym >> 4;
            const int ss = sym & 15;
            if (ss == 0 || ss > 10) { err = JEA_ERR_BAD_STREAM; break; }
            rd.refill();
            val = read_amplitude(rd, ss);
        }
        runs[n] = (uint16_t)run;
        values[n] = (int16_t)val;
        ++n;
        processed += 1 + run;
        // :707-710 goes on until the count is exactly 63 or an end-of-block entry comes; a count that steps over 63
        // keeps the reference reading entries without bound, which this kernel ends at the 64 entries its output holds
        if ((run == 0 && val == 0) || processed == 63) break;
        if (n == 64) { err = JEA_ERR_BAD_STREAM; break; }
    }
    if (rd.past_end()) err = JEA_ERR_BAD_STREAM;
    if (err) atomicExch(status, err);
    *dc_difference = diff;
    *n_entries = n;
    *end_bit = rd.pos * 8 - (uint64_t)rd.nbits;
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
void launch_huffman_encode_rle(hipStream_t s, int dc_difference, const uint16_t* d_runs, const int16_t* d_values, uint32_t n,
                               bool is_lum, const HuffTables* d_huff, uint32_t* d_bits, uint32_t* d_length, int32_t* d_status) {
    huffman_encode_rle_kernel<<<1, 64, 0, s>>>(dc_difference, d_runs, d_values, n, is_lum ? 1 : 0, d_huff, d_bits, d_length,
                                               d_status);
}
void launch_huffman_decode_rle(hipStream_t s, const uint8_t* d_stream, uint64_t len, uint64_t start_bit, bool is_lum,
                               const DecodeTables* d_tables, int32_t* d_dc_difference, uint16_t* d_runs, int16_t* d_values,
                               uint32_t* d_n_entries, uint64_t* d_end_bit, int32_t* d_status) {
    huffman_decode_rle_kernel<<<1, 64, 0, s>>>(d_stream, len, start_bit, is_lum ? 1 : 0, d_tables, d_dc_difference, d_runs,
                                               d_values, d_n_entries, d_end_bit, d_status);
}
void launch_huffman_decode_block(hipStream_t s, const uint8_t* d_stream, uint64_t len, uint64_t start_bit, bool is_lum,
                                 int last_dc, const DecodeTables* d_tables, int16_t* d_zz, uint64_t* d_end_bit,
                                 int32_t* d_last_dc, int32_t* d_status) {
    huffman_decode_block_kernel<<<1, 64, 0, s>>>(d_stream, len, start_bit, is_lum ? 1 : 0, last_dc, d_tables, d_zz,
                                                 d_end_bit, d_last_dc, d_status);
}

}  // namespace jea
