// Parallel Huffman entropy coding for gfx950.
//
// The reference appends the bits of one block-channel after another to a single
// serial BitStream (jpeg/src/encoder.cpp:21-27, jpeg/src/bit_stream.cpp:44-81),
// then byte-stuffs the committed bytes (bit_stream.cpp:127-134, intended
// semantics: a 0x00 after every 0xFF), zero-pads the last byte and appends EOI
// (encoder.cpp:29-32).  Here the same byte string is produced in parallel:
//
//   tile_bits_kernel    one thread per block-channel: run-length symbols, Huffman
//                       codes and amplitude bits (entropy_encoder.cpp:94-120,
//                       715-765) assembled MSB first into the block's own slot,
//                       plus the block's bit length; the DC difference needs only
//                       the previous MCU's DC of the same channel, read straight
//                       from the coefficients.  Then, per tile, the number of 0xFF
//                       bytes the tile would own for each of the eight bit offsets
//                       (mod 8) it may end up at
//   scan_tiles_ff_kernel  batches and pictures of more than 4096 tiles: exclusive scan
//                       of the per-tile bit totals; each tile's bit base picks its 0xFF
//                       count, scanned in the same pass (a single smaller picture skips
//                       this launch: every workgroup of the next kernel sums for itself)
//   pack_stuff_kernel   every tile shifts its blocks' bit strings to their bit
//                       offsets inside an LDS window and copies the bytes it owns to
//                       their final place, inserting 0x00 after 0xFF; header and EOI
//
// Stripes of one picture, whose bit offsets come from other contexts between two calls,
// use block_bits_kernel, scan_tiles_kernel, pack_kernel (into an unstuffed copy,
// counting its 0xFF bytes), scan_tiles_kernel and stuff_kernel instead.
//
// A tile is kTileBlocks = 252 consecutive block-channels (84 MCUs) of one image.
// Byte ownership: a byte belongs to the tile that holds its first bit, so the
// byte that straddles two tiles is produced by the earlier tile, which for that
// purpose also places the first kTileOverlap = 4 block-channels of the next tile: with
// the Annex K codes a block-channel has at least 4 bits, with per-image tables at
// least 2 (a one-bit DC code and a one-bit end-of-block code), so four of them are
// at least 8 bits, more than the 7 that can be missing.  No atomics on global
// memory and no zero-filled output are needed.
//
// Slots: the bit string of block j of a tile lives in words slot[w][j],
// w = 0..51 (1660 bits at most), of the tile's [52][256] word array, so that the
// 256 threads of a workgroup touch consecutive addresses for equal w.
#include <hip/hip_runtime.h>

#include "huffman_device.h"
#include <type_traits>

#include "kernels.h"
#include "tables.h"
#include "../../include/jpegenc_amd.h"

namespace jea {

namespace {

constexpr int kEntThreads = 256;
constexpr uint32_t kWinWords = 4096;  // 16 KiB LDS window of the packed stream
constexpr uint32_t kFoldTiles = 4096;  // single images up to this many tiles skip the scan launches

// Exclusive scan of one value per thread over a 256-thread workgroup.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* lds4, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    const uint32_t s0 = lds4[0], s1 = lds4[1], s2 = lds4[2], s3 = lds4[3];
    if (wave > 0) base += s0;
    if (wave > 1) base += s1;
    if (wave > 2) base += s2;
    total = s0 + s1 + s2 + s3;
    __syncthreads();
    return base + incl - v;
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {   // a value that is the same in every lane of the wave
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}

// The same scan with one barrier of the LDS-only kind; `lds4`: four words that the call before this one did not use
// (two sets, taken in turn).
__device__ __forceinline__ uint32_t block_exclusive_scan_256_lds(uint32_t v, uint32_t* lds4, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds4[wave] = incl;
    lds_barrier();
    uint32_t base = 0;
    const uint32_t s0 = lds4[0], s1 = lds4[1], s2 = lds4[2], s3 = lds4[3];
    if (wave > 0) base += s0;
    if (wave > 1) base += s1;
    if (wave > 2) base += s2;
    total = s0 + s1 + s2 + s3;
    return base + incl - v;
}

template <bool kEmit>
__global__ __launch_bounds__(kEntThreads) void block_bits_kernel(const int16_t* __restrict__ coefs, uint32_t nbc,
                                                                 uint32_t n_tiles,
                                                                 const HuffTables* __restrict__ huff,
                                                                 uint16_t* __restrict__ len16,
                                                                 uint32_t* __restrict__ tile_bits,
                                                                 uint32_t* __restrict__ slots,
                                                                 int32_t* __restrict__ status, StripeInfo st,
                                                                 const uint8_t* __restrict__ chunk_masks,
                                                                 ComponentMode mode,
                                                                 const int16_t* __restrict__ d_initial_dc = nullptr) {
    __shared__ std::conditional_t<kEmit, HuffTop, HuffLds> h;   // the writer's tables, or the plain codes for counting
    __shared__ uint32_t wsum[4];
    const uint32_t tile = blockIdx.x, img = blockIdx.y;
    const uint32_t b = tile * kTileBlocks + threadIdx.x;
    const bool has_block = threadIdx.x < kTileBlocks && b < nbc;
    // the block's own loads go out first, so that they are under way while the tables are copied
    BlockRegs blk;
    const int ch = (int)(b % 3u);
    int prev_dc = 0;
    if (has_block) {
        // previous DC: same channel, previous MCU; the chain starts at 0 (encoder.cpp:20) or, in a
        // stripe, at the last MCU of the stripe above
        const int16_t* p = coefs + ((size_t)img * nbc + b) * 64;
        load_block(blk, p, chunk_masks != nullptr ? (uint32_t)chunk_masks[(size_t)img * nbc + b] : 0xffu);
        // (d_initial_dc: the same three values in device memory, where another stripe's owner has put them)
        prev_dc = (b >= 3) ? (int)p[-192] : (d_initial_dc != nullptr ? (int)d_initial_dc[b] : (int)st.initial_dc[b]);
    }
    if constexpr (kEmit) load_huff_top<kEntThreads>(h, huff);
    else load_huff<kEntThreads>(h, huff);
    __syncthreads();
    uint32_t bits = 0;
    if (has_block) {
        bool ok;
        const int cls = mode.is_lum((uint32_t)ch) ? 0 : 1;   // Huffman table class (encoder.cpp:25)
        if constexpr (kEmit) {
            TopSink sink(slots + ((size_t)img * n_tiles + tile) * kSlotWords * kSlotStride, threadIdx.x);
            ok = encode_block_top<true>(blk, prev_dc, h.dc[cls], h.ac[cls], sink);
            sink.finish();
            bits = sink.bits();
        } else {
            CountSink sink;
            ok = encode_block(blk, prev_dc, h.dc[cls], h.ac[cls], sink);
            bits = sink.bits;
        }
        if (!ok) atomicExch(status, JEA_ERR_RANGE);
        len16[(size_t)img * nbc + b] = (uint16_t)bits;
    }
    uint32_t total;
    (void)block_exclusive_scan_256(bits, wsum, total);
    if (threadIdx.x == 0) tile_bits[(size_t)img * n_tiles + tile] = total;
}

// One workgroup per image: out[t] = sum of in[0..t-1], total -> img_total[img].
// Optionally (single image) also finishes the image-level bookkeeping that
// scan_images_kernel does for batches.
struct ImageLevel {
    int mode;               // 0 none, 1 unstuffed bases, 2 final offsets/sizes
    const uint64_t* img_bits;
    uint64_t* base_out;     // mode 1: img_ubase, mode 2: d_offsets
    uint64_t* size_out;     // mode 2: d_sizes
    uint64_t capacity;
    uint32_t fixed_bytes;   // mode 2: header + EOI bytes
    int32_t* status;
    // mode 2, a batch encoded in several passes: offset and size of the file before this pass's first one
    // (null: this pass's files start at offset 0)
    const uint64_t* prev_offset = nullptr;
    const uint64_t* prev_size = nullptr;
    __device__ uint64_t first_offset() const {
        return prev_offset != nullptr ? (*prev_offset + *prev_size + 63) / 64 * 64 : 0;
    }
};

__device__ __forceinline__ uint64_t unstuffed_bytes(uint64_t bits) { return (bits + 7) >> 3; }
__device__ __forceinline__ uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

__global__ __launch_bounds__(1024) void scan_tiles_kernel(const uint32_t* __restrict__ in, uint32_t n_tiles,
                                                          uint64_t* __restrict__ out,
                                                          uint64_t* __restrict__ img_total, ImageLevel lvl) {
    __shared__ uint64_t wave_sum[2][16];
    const uint32_t img = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 4096 tiles per step, four consecutive ones per thread: one 16-byte load and two 16-byte stores per thread, all
    // of a wave's next to one another, the next step's load asked for before this step's scan, and one barrier per
    // step (the waves' sums alternate between two buffers).  A stripe of 50 000 tiles took 76 us with one tile per
    // thread and step; a run of 49 consecutive tiles per thread, whose accesses do not coalesce, 100 us.
    const uint32_t* src = in + (size_t)img * n_tiles;
    uint64_t* dst = out + (size_t)img * n_tiles;
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15u) | (reinterpret_cast<uintptr_t>(dst) & 15u)) == 0;
    auto load4 = [&](uint32_t at) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (at < n_tiles) {
            if (aligned && at + 4u <= n_tiles) {
                q = *reinterpret_cast<const uint4*>(src + at);
            } else {
                q.x = src[at];
                if (at + 1u < n_tiles) q.y = src[at + 1u];
                if (at + 2u < n_tiles) q.z = src[at + 2u];
                if (at + 3u < n_tiles) q.w = src[at + 3u];
            }
        }
        return q;
    };
    uint64_t carry = 0;   // every thread keeps the running total itself
    uint4 next = load4(4u * threadIdx.x);
    uint32_t step = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 4096u, ++step) {
        const uint32_t at = t0 + 4u * threadIdx.x;
        const uint4 q = next;
        if (t0 + 4096u < n_tiles) next = load4(at + 4096u);
        const uint64_t v = (uint64_t)q.x + q.y + q.z + q.w;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        uint64_t* ws = wave_sum[step & 1u];
        if (lane == 63) ws[wave] = incl;
        __syncthreads();
        uint64_t base = carry, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint64_t x = ws[w];
            base += w < wave ? x : 0;
            all += x;
        }
        carry += all;
        if (at < n_tiles) {
            const uint64_t e0 = base + incl - v, e1 = e0 + q.x, e2 = e1 + q.y, e3 = e2 + q.z;
            if (aligned && at + 4u <= n_tiles) {
                *reinterpret_cast<ulonglong2*>(dst + at) = make_ulonglong2(e0, e1);
                *reinterpret_cast<ulonglong2*>(dst + at + 2) = make_ulonglong2(e2, e3);
            } else {
                dst[at] = e0;
                if (at + 1u < n_tiles) dst[at + 1u] = e1;
                if (at + 2u < n_tiles) dst[at + 2u] = e2;
                if (at + 3u < n_tiles) dst[at + 3u] = e3;
            }
        }
    }
    __shared__ uint64_t carry_s;
    if (threadIdx.x == 0) carry_s = carry;
    if (threadIdx.x == 0) {
        const uint64_t total = carry_s;
        img_total[img] = total;
        if (gridDim.x == 1 && lvl.mode == 1) {
            lvl.base_out[0] = 0;
            if (align_up(unstuffed_bytes(total), 16) > lvl.capacity) atomicExch(lvl.status, JEA_ERR_CAPACITY);
        } else if (gridDim.x == 1 && lvl.mode == 2) {
            const uint64_t size = unstuffed_bytes(lvl.img_bits[0]) + total + lvl.fixed_bytes;
            lvl.base_out[0] = 0;
            lvl.size_out[0] = size;
            if (size > lvl.capacity) atomicExch(lvl.status, JEA_ERR_CAPACITY);
        }
    }
}

// Batches: exclusive scan over images of the per-image byte sizes, single workgroup.
// mode 1: size = align16(unstuffed bytes)                 (img_total = img_bits)
// mode 2: size = unstuffed bytes + 0xFF count + fixed      (img_total = img_ff), placed at 64-byte steps
__global__ __launch_bounds__(1024) void scan_images_kernel(uint32_t n_images, const uint64_t* __restrict__ img_total,
                                                           ImageLevel lvl) {
    __shared__ uint64_t wave_sum[16];
    __shared__ uint64_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = lvl.mode == 2 ? lvl.first_offset() : 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_images; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        uint64_t size = 0, v = 0;
        if (i < n_images) {
            if (lvl.mode == 1) {
                v = align_up(unstuffed_bytes(img_total[i]), 16);
            } else {
                size = unstuffed_bytes(lvl.img_bits[i]) + img_total[i] + lvl.fixed_bytes;
                v = align_up(size, 64);
            }
        }
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint64_t base = carry_s;
        for (int w = 0; w < wave; ++w) base += wave_sum[w];
        const uint64_t excl = base + incl - v;
        if (i < n_images) {
            lvl.base_out[i] = excl;
            if (lvl.mode == 2) {
                lvl.size_out[i] = size;
                if (excl + size > lvl.capacity) atomicExch(lvl.status, JEA_ERR_CAPACITY);
            } else if (excl + v > lvl.capacity) {
                atomicExch(lvl.status, JEA_ERR_CAPACITY);
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = base + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kEntThreads) void pack_kernel(uint32_t nbc, uint32_t n_tiles,
                                                           const uint32_t* __restrict__ slots,
                                                           const uint16_t* __restrict__ len16,
                                                           const uint32_t* __restrict__ tile_bits,
                                                           const uint64_t* __restrict__ tile_base,
                                                           const uint64_t* __restrict__ img_bits,
                                                           const uint64_t* __restrict__ img_ubase,
                                                           uint8_t* __restrict__ unstuffed,
                                                           uint64_t unstuffed_capacity,
                                                           uint32_t* __restrict__ tile_ff,
                                                           int32_t* __restrict__ status, StripeInfo st,
                                                           const StripeInfo* __restrict__ d_st = nullptr) {
    __shared__ uint32_t win[kWinWords + 2];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t ff_count;
    if (d_st != nullptr) st = *d_st;   // the stripe's place in the stream, worked out on the device (stripe_place_kernel)
    const uint32_t tile = blockIdx.x, img = blockIdx.y;
    const int32_t status_now = *status;                       // looked at after the other loads are under way
    const uint32_t b = tile * kTileBlocks + threadIdx.x;
    // 252 own block-channels + the first four of the next tile (threads 252..255)
    const bool has_block = threadIdx.x < kTileBlocks + kTileOverlap && b < nbc;
    // last tile of a stripe that has a successor: one thread stands in for the
    // successor's first blocks with the 32 head bits it was given
    const bool head_thread = st.has_next && tile + 1 == n_tiles && threadIdx.x == kTileBlocks;
    const uint32_t my_len = has_block ? len16[(size_t)img * nbc + b] : (head_thread ? 32u : 0u);
    if (status_now != 0) return;
    if (threadIdx.x == 0) ff_count = 0;
    uint32_t tile_total;
    const uint32_t local_off = block_exclusive_scan_256(my_len, wsum, tile_total);  // has barriers

    const uint64_t* tb = tile_base + (size_t)img * n_tiles;
    uint64_t base_bits = tb[tile];
    uint64_t end_bits = (tile + 1 == n_tiles) ? img_bits[img] : tb[tile + 1];
    base_bits += st.bit_base;  // image-stream coordinates
    end_bits += st.bit_base;
    const uint64_t b0 = (base_bits + 7) >> 3, b1 = (end_bits + 7) >> 3;  // owned bytes [b0, b1)
    const uint64_t d0 = base_bits >> 5;                      // image-relative dword of window word 0
    const uint32_t rel_start = (uint32_t)(base_bits & 31) + local_off;
    const uint32_t n_words = (uint32_t)((b1 * 8 - d0 * 32 + 31) >> 5);  // words that hold owned bytes
    const uint64_t ubase = img_ubase[img];   // the scans have checked the capacity
    uint8_t* dst = unstuffed + ubase - st.origin_byte;  // indexed by image-stream byte

    // this thread's slot: own tile, or column 0..3 of the next tile for the four extra threads
    const uint32_t slot_tile = threadIdx.x < kTileBlocks ? tile : tile + 1;
    const uint32_t slot_col = threadIdx.x < kTileBlocks ? threadIdx.x : threadIdx.x - kTileBlocks;
    const uint32_t* slot = slots + ((size_t)img * n_tiles + slot_tile) * kSlotWords * kSlotStride + slot_col;
    const uint32_t my_words = (my_len + 31) >> 5;
    const uint32_t sh = rel_start & 31u;

    uint32_t my_ff = 0;
    for (uint32_t wlo = 0; wlo < n_words; wlo += kWinWords) {
        const uint32_t words_here = (n_words - wlo < kWinWords) ? (n_words - wlo) : kWinWords;
        for (uint32_t i = threadIdx.x; i < words_here + 2; i += kEntThreads) win[i] = 0;
        __syncthreads();
        // word w of the block's string covers stream words first+w and first+w+1
        const uint32_t first = rel_start >> 5;
        for (uint32_t w = 0; w < my_words; ++w) {
            const uint32_t idx = first + w - wlo;        // wraps below the window: rejected by the range test
            if (idx + 1 < kWinWords + 1 || idx == 0xffffffffu) {
                const uint32_t v = head_thread ? st.next_head : slot[(size_t)w * kSlotStride];
                if (idx < kWinWords) atomicOr(&win[idx], v >> sh);
                if (sh != 0 && idx + 1 < kWinWords) atomicOr(&win[idx + 1], v << (32u - sh));
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < words_here; i += kEntThreads) {
            const uint32_t v = win[i];
            const uint64_t byte0 = (d0 + wlo + i) * 4;  // image-relative index of the word's first byte
            const uint32_t bytes[4] = {v >> 24, (v >> 16) & 0xffu, (v >> 8) & 0xffu, v & 0xffu};
            if (byte0 >= b0 && byte0 + 4 <= b1) {
                *reinterpret_cast<uint32_t*>(dst + byte0) = __builtin_bswap32(v);
#pragma unroll
                for (int k = 0; k < 4; ++k) my_ff += (bytes[k] == 0xffu);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint64_t by = byte0 + k;
                    if (by >= b0 && by < b1) {
                        dst[by] = (uint8_t)bytes[k];
                        my_ff += (bytes[k] == 0xffu);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (my_ff) atomicAdd(&ff_count, my_ff);
    __syncthreads();
    if (threadIdx.x == 0) tile_ff[(size_t)img * n_tiles + tile] = ff_count;
}

__global__ __launch_bounds__(kEntThreads) void stuff_kernel(uint32_t n_tiles, const uint32_t* __restrict__ tile_bits,
                                                            const uint64_t* __restrict__ tile_base,
                                                            const uint64_t* __restrict__ img_bits,
                                                            const uint64_t* __restrict__ img_ubase,
                                                            const uint8_t* __restrict__ unstuffed,
                                                            const uint32_t* __restrict__ tile_ff,
                                                            const uint64_t* __restrict__ tile_ffbase,
                                                            const uint64_t* __restrict__ img_ff,
                                                            uint64_t* __restrict__ offsets,
                                                            uint64_t* __restrict__ sizes, HeaderBytes header,
                                                            uint32_t header_len, uint32_t with_eoi,
                                                            uint8_t* __restrict__ out, uint64_t out_capacity,
                                                            int32_t* __restrict__ status, StripeInfo st,
                                                            const StripeInfo* __restrict__ d_st = nullptr) {
    __shared__ uint32_t wsum[4];
    if (d_st != nullptr) st = *d_st;
    const uint32_t tile = blockIdx.x, img = blockIdx.y;
    if (*status != 0) return;
    const uint64_t* tb = tile_base + (size_t)img * n_tiles;
    const uint64_t base_bits = tb[tile] + st.bit_base;
    const uint64_t end_bits = ((tile + 1 == n_tiles) ? img_bits[img] : tb[tile + 1]) + st.bit_base;
    const uint64_t b0 = (base_bits + 7) >> 3, b1 = (end_bits + 7) >> 3;
    const uint64_t ff_before = tile_ffbase[(size_t)img * n_tiles + tile];
    const uint64_t file_offset = offsets[img];
    if (st.active) {
        // a stripe's segment: its owned bytes start at st.first_byte; exact size known here
        const uint64_t tail = (tile + 1 == n_tiles && with_eoi) ? 2 : 0;
        if (header_len + (b1 - st.first_byte) + ff_before + tile_ff[(size_t)img * n_tiles + tile] + tail > out_capacity) {
            if (threadIdx.x == 0) atomicExch(status, JEA_ERR_CAPACITY);
            return;
        }
        if (tile + 1 == n_tiles && threadIdx.x == 0)
            sizes[img] = header_len + (b1 - st.first_byte) + img_ff[img] + tail;
    }
    const uint8_t* src = unstuffed + img_ubase[img] - st.origin_byte;  // indexed by image-stream byte
    uint8_t* file = out + file_offset;
    // + image-stream byte index + 0xFFs before it inside the tile
    uint8_t* dst = file + header_len + ff_before - st.first_byte;

    if (tile == 0)
        for (uint32_t i = threadIdx.x; i < header_len; i += kEntThreads) file[i] = header.b[i];
    if (tile + 1 == n_tiles && with_eoi && threadIdx.x == 0) {
        uint8_t* e = file + header_len + (b1 - st.first_byte) + img_ff[img];
        e[0] = 0xFF;
        e[1] = 0xD9;
    }

    uint64_t carry = 0;  // 0xFF bytes of this tile before the current chunk
    // four bytes per thread and step: the bytes one store instruction of a wave writes lie close together
    constexpr int kBytes = 4;
    for (uint64_t c0 = b0; c0 < b1; c0 += (uint64_t)kEntThreads * kBytes) {
        const uint64_t i0 = c0 + (uint64_t)threadIdx.x * kBytes;
        uint8_t v[kBytes];
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < kBytes; ++k) {
            v[k] = (i0 + k < b1) ? src[i0 + k] : 0;
            cnt += (v[k] == 0xFF);
        }
        uint32_t total;
        uint64_t pos = carry + block_exclusive_scan_256(cnt, wsum, total);
#pragma unroll
        for (int k = 0; k < kBytes; ++k) {
            if (i0 + k < b1) {
                dst[i0 + k + pos] = v[k];
                if (v[k] == 0xFF) {
                    ++pos;
                    dst[i0 + k + pos] = 0x00;
                }
            }
        }
        carry += total;
    }
}

// ---------------------------------------------------------------------------
// Two launches instead of three (plus scans).  Where a tile's bytes go in the file depends on the 0xFF bytes
// before it, and which bytes are 0xFF depends on how the tile's bits fall on the byte grid: on the bit total
// of the tiles before it, modulo 8.  So the first kernel, besides writing the blocks' bit strings, counts for
// each of the eight offsets the tile may end up at the 0xFF bytes it would own; the second kernel (or the
// scan in between) gets the 0xFF prefix from those counts along with the bit totals, and stuffs straight out
// of its LDS window.  No unstuffed copy in memory.
// ---------------------------------------------------------------------------

// ORs every thread's string (my_len bits in slot words, starting `start` bits into the window's word 0 of
// window pass wlo) into win[0 .. kWinWords]; word kWinWords is complete too, as the successor of the last.
__device__ __forceinline__ void or_strings(uint32_t* win, const uint32_t* __restrict__ slot, uint32_t my_len,
                                           uint32_t start, uint32_t wlo) {
    const uint32_t my_words = (my_len + 31) >> 5, sh = start & 31u, first = start >> 5;
    // Eight loads under way at a time: as one loop, every word's load is waited for before the next is issued,
    // and a string of noise (a dozen words) then costs a dozen trips to memory, one after the other.
    constexpr uint32_t kBatch = 8;
    for (uint32_t w0 = 0; w0 < my_words; w0 += kBatch) {
        uint32_t v[kBatch];
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t w = w0 + k, idx = first + w - wlo;   // wraps below the window: rejected by the range tests
            v[k] = (w < my_words && (idx <= kWinWords || idx == 0xffffffffu)) ? slot[(size_t)w * kSlotStride] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kBatch; ++k) {
            const uint32_t w = w0 + k, idx = first + w - wlo;
            if (w < my_words && (idx <= kWinWords || idx == 0xffffffffu)) {
                if (idx <= kWinWords) atomicOr(&win[idx], v[k] >> sh);
                if (sh != 0 && idx + 1 <= kWinWords) atomicOr(&win[idx + 1], v[k] << (32u - sh));
            }
        }
    }
}

template <bool kPlanar>
__global__ __launch_bounds__(kEntThreads, 8) void tile_bits_kernel(const int16_t* __restrict__ coefs, uint32_t nbc,
                                                                uint32_t n_tiles,
                                                                const HuffTables* __restrict__ huff,
                                                                const uint8_t* __restrict__ chunk_masks,
                                                                ComponentMode mode, uint16_t* __restrict__ len16,
                                                                uint32_t* __restrict__ slots,
                                                                uint32_t* __restrict__ tile_bits,
                                                                uint4* __restrict__ tile_ff8,
                                                                int32_t* __restrict__ status, uint32_t tables_per_image) {
    __shared__ HuffTop h;
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t count[8];
    __shared__ uint32_t lens[kEntThreads];
    const uint32_t tile = blockIdx.x, img = blockIdx.y;
    const uint32_t b = tile * kTileBlocks + threadIdx.x;
    // 252 own block-channels, and the first four of the next tile once more (threads 252..255): the byte that
    // straddles two tiles belongs to the earlier one, whose 0xFF count needs those bits
    const bool has_block = threadIdx.x < kTileBlocks + kTileOverlap && b < nbc;
    BlockRegs blk;
    const int ch = (int)(b % 3u);
    int prev_dc = 0;
    if (has_block) {
        if (kPlanar) {
            const uint32_t n_mcu = nbc / 3u, mcu = b / 3u;
            const uint4* q = reinterpret_cast<const uint4*>(coefs) + ((size_t)img * 24 + ch) * n_mcu + mcu;
            load_block_planar(blk, q, (size_t)3 * n_mcu, (uint32_t)chunk_masks[(size_t)img * nbc + b]);
            prev_dc = (b >= 3) ? (int)*reinterpret_cast<const int16_t*>(q - 1) : 0;
        } else {
            const int16_t* p = coefs + ((size_t)img * nbc + b) * 64;
            load_block(blk, p, chunk_masks != nullptr ? (uint32_t)chunk_masks[(size_t)img * nbc + b] : 0xffu);
            prev_dc = (b >= 3) ? (int)p[-192] : 0;
        }
    }
    load_huff_top<kEntThreads>(h, huff + (tables_per_image ? blockIdx.y : 0u));
    if (threadIdx.x < 8) count[threadIdx.x] = 0;
    __syncthreads();

    uint32_t* const tile_slots = slots + ((size_t)img * n_tiles + tile) * kSlotWords * kSlotStride;
    uint32_t* const slot = tile_slots + threadIdx.x;
    uint32_t my_len = 0;
    if (has_block) {
        const int cls = mode.is_lum((uint32_t)ch) ? 0 : 1;   // Huffman table class (encoder.cpp:25)
        TopSink sink(tile_slots, threadIdx.x);
        // the chunk-planar layout is the library's own: coefficients from its quantiser
        if (!encode_block_top<!kPlanar>(blk, prev_dc, h.dc[cls], h.ac[cls], sink)) atomicExch(status, JEA_ERR_RANGE);
        sink.finish();
        my_len = sink.bits();
        if (threadIdx.x < kTileBlocks) len16[(size_t)img * nbc + b] = (uint16_t)my_len;
    }
    lens[threadIdx.x] = my_len;
    uint32_t with_next;
    const uint32_t local_off = block_exclusive_scan_256(my_len, wsum, with_next);   // has barriers
    if (threadIdx.x == kTileBlocks) tile_bits[(size_t)img * n_tiles + tile] = local_off;   // its prefix is the total of the tile's own blocks

    // Bit p of the tile's string starts a 0xFF byte for the alignments that put a byte boundary there, if bits
    // p .. p+7 are ones; the tile owns the bytes that start inside it.  Each thread reads its string back, with
    // the seven bits that follow it: those are in the next two strings (a block's string has at least four
    // bits; past the picture's end the padding is zeros).  Tallies go from the string's positions to the tile's.
    if (threadIdx.x < kTileBlocks && has_block) {
        const uint32_t len1 = lens[threadIdx.x + 1], len2 = lens[threadIdx.x + 2];
        const uint32_t head1 = len1 ? slot[1] : 0u, head2 = (len1 < 32 && len2) ? slot[2] : 0u;
        const uint32_t following = head1 | (len1 < 32 ? head2 >> len1 : 0u);
        // The string's words as the stream has them: the partial last word filled up with the following bits, and the
        // rest of those after it.  Every place of the words before the last counts, of the last those of its own bits.
        const uint32_t rem = my_len & 31u, last = (my_len - 1) >> 5;   // a string has at least four bits
        const uint32_t fill = rem ? following >> rem : 0u;
        const uint32_t after = rem ? following << (32u - rem) : following;
        FfTally tally;
        uint32_t cur = slot[0] | (last == 0 ? fill : 0u);
        uint32_t batches = 0;
        for (uint32_t i0 = 0; i0 < last; i0 += 4) {   // four loads under way at a time
            uint32_t w[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                w[k] = (i0 + k + 1 <= last) ? slot[(size_t)(i0 + k + 1) * kSlotStride] : 0u;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t i = i0 + k;
                if (i < last) {
                    const uint32_t next = w[k] | (i + 1 == last ? fill : 0u);
                    const uint32_t m = ff_starts(cur, next);
                    if (m != 0) tally.add(m);
                    cur = next;
                }
            }
            if (++batches == 3) {   // twelve words: the tally's fields hold fifteen
                tally.flush();
                batches = 0;
            }
        }
        const uint32_t m_last = ff_starts(cur, after) & (rem ? ~0u << (32u - rem) : ~0u);
        if (m_last != 0) tally.add(m_last);
        tally.flush();
        uint32_t any = 0;
#pragma unroll
        for (uint32_t r = 0; r < 8; ++r) any |= tally.total[r];
        if (any != 0) {
            const uint32_t shift = local_off & 7u;
#pragma unroll
            for (uint32_t r = 0; r < 8; ++r)
                if (tally.total[r]) atomicAdd(&count[(r + shift) & 7u], tally.total[r]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)   // eight 16-bit counts: a tile has at most 252 * 1660 / 8 bytes
        tile_ff8[(size_t)img * n_tiles + tile] = make_uint4(count[0] | count[1] << 16, count[2] | count[3] << 16, count[4] | count[5] << 16,
                                    count[6] | count[7] << 16);
}

// The count for tiles whose bit base is `base_bits`: byte boundaries fall on the tile's bits p = -base mod 8.
__device__ __forceinline__ uint32_t ff_at(const uint4& f, uint32_t base_bits) {
    const uint32_t r = (8u - (base_bits & 7u)) & 7u;
    // by shifts: a choice among the four words turns into an indexed load from a copy in scratch memory
    const uint64_t lo = (uint64_t)f.x | (uint64_t)f.y << 32, hi = (uint64_t)f.z | (uint64_t)f.w << 32;
    return (uint32_t)(((r & 4u) ? hi : lo) >> (16u * (r & 3u))) & 0xffffu;
}

// One workgroup per image.  Exclusive scan of the tiles' bit totals; each tile's bit base picks its 0xFF count
// out of the eight, and those are scanned in the same pass.  A single image's offset and size are final here.
__global__ __launch_bounds__(1024) void scan_tiles_ff_kernel(const uint32_t* __restrict__ tile_bits,
                                                             const uint4* __restrict__ tile_ff8, uint32_t n_tiles,
                                                             uint64_t* __restrict__ tile_base,
                                                             uint64_t* __restrict__ tile_ffbase,
                                                             uint64_t* __restrict__ img_bits,
                                                             uint64_t* __restrict__ img_ff, ImageLevel lvl) {
    __shared__ uint64_t wave_bits[16], wave_ff[16];
    __shared__ uint64_t carry_bits, carry_ff;
    const uint32_t img = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_bits = carry_ff = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const size_t at = (size_t)img * n_tiles + t;
        const uint64_t v = (t < n_tiles) ? (uint64_t)tile_bits[at] : 0;
        const uint4 f = (t < n_tiles) ? tile_ff8[at] : make_uint4(0, 0, 0, 0);
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) wave_bits[wave] = incl;
        __syncthreads();
        uint64_t base = carry_bits;
        for (int w = 0; w < wave; ++w) base += wave_bits[w];
        const uint64_t bit_base = base + incl - v;
        const uint64_t ff = (t < n_tiles) ? ff_at(f, (uint32_t)bit_base) : 0;
        uint64_t incl_ff = ff;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t u = __shfl_up(incl_ff, d);
            if (lane >= d) incl_ff += u;
        }
        if (lane == 63) wave_ff[wave] = incl_ff;
        __syncthreads();
        uint64_t base_ff = carry_ff;
        for (int w = 0; w < wave; ++w) base_ff += wave_ff[w];
        if (t < n_tiles) {
            tile_base[at] = bit_base;
            tile_ffbase[at] = base_ff + incl_ff - ff;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry_bits = base + incl;
            carry_ff = base_ff + incl_ff;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        img_bits[img] = carry_bits;
        img_ff[img] = carry_ff;
        if (gridDim.x == 1) {
            const uint64_t size = unstuffed_bytes(carry_bits) + carry_ff + lvl.fixed_bytes;
            const uint64_t at = lvl.first_offset();
            lvl.base_out[0] = at;
            lvl.size_out[0] = size;
            if (at + size > lvl.capacity) atomicExch(lvl.status, JEA_ERR_CAPACITY);
        }
    }
}

// Stuffs the owned bytes [b0, b1) that lie in a window of the packed stream (win[0 .. words_here), its first
// byte is stream byte win_byte) to dst + stream byte index + 0xFFs of this tile before it: bit_stream.cpp:127-134
// (intended semantics) as count -> scan over the workgroup -> scatter, kWords words per thread and step.
// Short windows (a tile of a smooth picture has about a hundred words) take one word per thread, so that the
// step's instructions are not spent on threads without bytes; long ones two: the fewer bytes a thread has,
// the closer together lie the bytes that one store instruction of the wave writes (with four words per
// thread a 4K picture of noise encoded at 134 instead of 141 GPix/s).
// The window's bytes [own_from, own_to) are the owned ones (own_to at most 4 * words_here); `out0`: where the window's
// byte 0 goes if no 0xFF of this tile precedes it -- the same for the whole workgroup and handed to the stores as
// such (a scalar base and a 32-bit offset per lane: one sum per store instead of two 64-bit ones); carry: the
// tile's 0xFF bytes so far.
template <uint32_t kWords>
__device__ __forceinline__ void stuff_window(const uint32_t* win, uint32_t words_here, uint32_t own_from, uint32_t own_to,
                                             uint8_t* out0, uint32_t& carry, uint32_t* wsum8) {
    uint32_t turn = 0;
    for (uint32_t w0 = 0; w0 < words_here; w0 += kEntThreads * kWords, turn ^= 4u) {
        const uint32_t wi = w0 + threadIdx.x * kWords;
        uint32_t bytes[4 * kWords];
        uint32_t owned = 0, cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < kWords; ++j) {
            const uint32_t word = wi + j;
            const uint32_t v = word < words_here ? win[word] : 0u;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t k = 4 * j + q, x = 4 * word + q;   // window byte index
                bytes[k] = (v >> (24 - 8 * q)) & 0xffu;
                if (x >= own_from && x < own_to) {
                    owned |= 1u << k;
                    cnt += bytes[k] == 0xffu;
                }
            }
        }
        uint32_t total;
        uint32_t pos = block_exclusive_scan_256_lds(cnt, wsum8 + turn, total);   // the stores of the turn before stay under way
        pos += 4 * wi + carry;
#pragma unroll
        for (uint32_t k = 0; k < 4 * kWords; ++k) {
            if ((owned >> k) & 1u) {
                out0[k + pos] = (uint8_t)bytes[k];
                if (bytes[k] == 0xffu) {
                    ++pos;
                    out0[k + pos] = 0x00;
                }
            }
        }
        carry += (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
    }
}

// kFold: one image of at most kFoldTiles tiles, every workgroup works out the sums over the tiles before it.
// Otherwise the sums come from scan_tiles_ff_kernel (and scan_images_kernel), which have also written the
// files' offsets and sizes and checked the capacity.
template <bool kFold>
__global__ __launch_bounds__(kEntThreads, 7) void pack_stuff_kernel(uint32_t nbc, uint32_t n_tiles,
                                                                    const uint32_t* __restrict__ slots,
                                                                    const uint16_t* __restrict__ len16,
                                                                    const uint32_t* __restrict__ tile_bits,
                                                                    const uint4* __restrict__ tile_ff8,
                                                                    const uint64_t* __restrict__ tile_base,
                                                                    const uint64_t* __restrict__ tile_ffbase,
                                                                    uint64_t* __restrict__ offsets,
                                                                    uint64_t* __restrict__ sizes, HeaderBytes header,
                                                                    uint32_t header_len, uint32_t with_eoi,
                                                                    uint8_t* __restrict__ out, uint64_t out_capacity,
                                                                    int32_t* __restrict__ status) {
    __shared__ uint32_t win[kWinWords + 2];
    __shared__ uint32_t wsum[8], psum[8], fsum[4];
    const uint32_t tile = blockIdx.x, img = blockIdx.y;
    const int32_t status_now = *status;
    const uint32_t b = tile * kTileBlocks + threadIdx.x;
    const bool has_block = threadIdx.x < kTileBlocks + kTileOverlap && b < nbc;
    uint64_t base_bits, ff_before;
    uint32_t own_bits, own_ff, my_len, local_off;
    if (kFold) {
        // Tiles before this one, a contiguous run of them per thread: their bit totals give the bit base of each
        // (after one scan over the threads), its alignment, and with that its 0xFF count.
        const uint32_t per_thread = (tile + kEntThreads - 1) / kEntThreads;
        const uint32_t j0 = threadIdx.x * per_thread, j1 = (j0 + per_thread < tile) ? j0 + per_thread : tile;
        // Up to kRun tiles per thread (any picture up to 1792 tiles; a 3840x2160 one has 1543): everything is loaded
        // before the scan, all eight counts of each tile, so that no load waits for the scan's result.
        constexpr uint32_t kRun = 7;
        const bool short_run = per_thread <= kRun;
        uint32_t run_bits = 0, rb[kRun];
        uint4 rf[kRun];
        if (short_run) {
#pragma unroll
            for (uint32_t u = 0; u < kRun; ++u) {
                const bool in = j0 + u < j1;
                rb[u] = in ? tile_bits[j0 + u] : 0u;
                rf[u] = in ? tile_ff8[j0 + u] : make_uint4(0, 0, 0, 0);
                run_bits += rb[u];
            }
        } else {
            for (uint32_t j = j0; j < j1; ++j) run_bits += tile_bits[j];
        }
        own_bits = tile_bits[tile];
        const uint4 own_ff8 = tile_ff8[tile];
        my_len = has_block ? len16[b] : 0u;
        // An error of an earlier launch ends this one.  JEA_ERR_CAPACITY is what other workgroups of this very
        // launch may be raising right now: the waves of a workgroup could see it differently, and an early exit
        // of some of them would leave the sums below incomplete -- so that one is left to the per-tile check.
        if (status_now != 0 && status_now != JEA_ERR_CAPACITY) return;
        // one scan for both: the runs' bit totals (a) and the lengths of this tile's strings (b)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t incl_a = run_bits, incl_b = my_len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t ta = __shfl_up(incl_a, d), tb = __shfl_up(incl_b, d);
            if (lane >= d) {
                incl_a += ta;
                incl_b += tb;
            }
        }
        if (lane == 63) {
            psum[wave] = incl_a;
            psum[4 + wave] = incl_b;
        }
        __syncthreads();
        uint32_t base_a = 0, base_b = 0, bits_before = 0;   // fits: at most 4096 tiles of 252 * 1660 bits
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) {
                base_a += psum[w];
                base_b += psum[4 + w];
            }
            bits_before += psum[w];
        }
        uint32_t running = base_a + incl_a - run_bits;
        local_off = base_b + incl_b - my_len;
        uint32_t ff_run = 0;
        if (short_run) {
#pragma unroll
            for (uint32_t u = 0; u < kRun; ++u) {
                ff_run += ff_at(rf[u], running);
                running += rb[u];
            }
        } else {
            for (uint32_t j = j0; j < j1; ++j) {
                ff_run += ff_at(tile_ff8[j], running);
                running += tile_bits[j];
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ff_run += __shfl_xor(ff_run, d);
        if (lane == 0) fsum[wave] = ff_run;
        __syncthreads();
        ff_before = (uint64_t)fsum[0] + fsum[1] + fsum[2] + fsum[3];
        base_bits = bits_before;
        own_ff = ff_at(own_ff8, bits_before);
    } else {
        const size_t t = (size_t)img * n_tiles + tile;
        base_bits = tile_base[t];
        ff_before = tile_ffbase[t];
        own_bits = tile_bits[t];
        const uint4 own_ff8 = tile_ff8[t];
        my_len = has_block ? len16[(size_t)img * nbc + b] : 0u;
        // An error ends the workgroup -- all of it or none of it.  The status word may be raised while this kernel
        // runs (by a neighbouring piece of a batch that is encoded on two streams), so the waves of a workgroup can
        // read different values; each leaves what it saw in LDS, the scan's barriers lie between that store and the
        // reads, and every wave then takes the same decision.
        if ((threadIdx.x & 63u) == 0) psum[threadIdx.x >> 6] = (uint32_t)status_now;
        uint32_t tile_total;
        local_off = block_exclusive_scan_256(my_len, wsum, tile_total);
        if ((psum[0] | psum[1] | psum[2] | psum[3]) != 0u) return;
        own_ff = ff_at(own_ff8, (uint32_t)base_bits);
    }
    // The same in every lane, but out of sums that each lane made for itself: as scalars again, so that what follows
    // from them -- the window's place in the file above all -- is worked out once, not per lane.
    base_bits = uniform64(base_bits);
    ff_before = uniform64(ff_before);
    const uint64_t end_bits = base_bits + own_bits;
    const uint64_t b0 = (base_bits + 7) >> 3, b1 = (end_bits + 7) >> 3;  // owned bytes [b0, b1) of the unstuffed stream
    uint8_t* file = out;
    if (kFold) {
        const uint64_t tail = (tile + 1 == n_tiles && with_eoi) ? 2 : 0;
        if (header_len + b1 + ff_before + own_ff + tail > out_capacity) {
            if (threadIdx.x == 0) atomicExch(status, JEA_ERR_CAPACITY);
            return;
        }
        if (tile + 1 == n_tiles && threadIdx.x == 0) {
            offsets[0] = 0;
            sizes[0] = header_len + b1 + ff_before + own_ff + tail;
        }
    } else {
        file += offsets[img];
    }
    if (tile == 0) {
        // three fixed steps cover the 608 bytes; as a loop over header_len the compiler unrolls it eightfold, and
        // this code, which one workgroup runs, pushes the kernel over its register limit
#pragma unroll
        for (uint32_t r = 0; r < (sizeof(header.b) + kEntThreads - 1) / kEntThreads; ++r) {
            const uint32_t i = r * kEntThreads + threadIdx.x;
            if (i < header_len) file[i] = header.b[i < sizeof(header.b) ? i : 0];
        }
    }
    if (tile + 1 == n_tiles && with_eoi && threadIdx.x == 0) {
        uint8_t* e = file + header_len + b1 + ff_before + own_ff;
        e[0] = 0xFF;
        e[1] = 0xD9;
    }

    const uint64_t d0 = base_bits >> 5;                      // image-relative dword of window word 0
    const uint32_t rel_start = (uint32_t)(base_bits & 31) + local_off;
    const uint32_t n_words = (uint32_t)((b1 * 8 - d0 * 32 + 31) >> 5);  // words that hold owned bytes
    // this thread's slot: own tile, or column 0..3 of the next tile for the four extra threads
    const uint32_t slot_tile = threadIdx.x < kTileBlocks ? tile : tile + 1;
    const uint32_t slot_col = threadIdx.x < kTileBlocks ? threadIdx.x : threadIdx.x - kTileBlocks;
    const uint32_t* slot = slots + ((size_t)img * n_tiles + slot_tile) * kSlotWords * kSlotStride + slot_col;
    uint8_t* const dst = file + header_len + ff_before;      // + stream byte index + 0xFFs of this tile before it
    uint32_t carry = 0;   // at most 252 * 1660 / 8 bytes a tile
    for (uint32_t wlo = 0; wlo < n_words; wlo += kWinWords) {
        const uint32_t words_here = (n_words - wlo < kWinWords) ? (n_words - wlo) : kWinWords;
        for (uint32_t i = threadIdx.x; i < words_here + 1; i += kEntThreads) win[i] = 0;
        __syncthreads();
        or_strings(win, slot, my_len, rel_start, wlo);
        __syncthreads();
        // owned bytes relative to the window: b0 lies in its first word or before the later windows, b1 at or after its start
        const uint64_t win_byte = (d0 + wlo) * 4;
        const uint32_t own_from = b0 > win_byte ? (uint32_t)(b0 - win_byte) : 0u;
        const uint64_t left = b1 > win_byte ? b1 - win_byte : 0;
        const uint32_t own_to = left < 4ull * words_here ? (uint32_t)left : 4u * words_here;
        if (words_here <= kEntThreads)
            stuff_window<1>(win, words_here, own_from, own_to, dst + win_byte, carry, wsum);
        else
            stuff_window<2>(win, words_here, own_from, own_to, dst + win_byte, carry, wsum);
    }
}

// ---------------------------------------------------------------------------
// Per-image Huffman tables: symbol counts, then the tables, then (after the files are written) the table
// bytes into the headers.  The procedure is this build's own (the reference has none) and is stated once, in
// the CPU restatement's jo_optimal_table; what follows computes the same, step for step.
// ---------------------------------------------------------------------------

template <bool kPlanar>
__global__ __launch_bounds__(kEntThreads) void symbol_histogram_kernel(const int16_t* __restrict__ coefs, uint32_t nbc,
                                                                       uint32_t n_tiles,
                                                                       const uint8_t* __restrict__ chunk_masks,
                                                                       ComponentMode mode, uint32_t* __restrict__ hist) {
    // [copy][class][DC symbols at 0, AC symbols at 256]: most blocks of a smooth picture count the same two or three
    // symbols, and atomics of one wave on one address go one after the other; eight copies, chosen by lane,
    // cut those queues to an eighth
    constexpr uint32_t kCopies = 8;
    __shared__ uint32_t hc[kCopies][2 * 512];
    uint32_t* const h = hc[threadIdx.x & (kCopies - 1)];
    const uint32_t img = blockIdx.y;
    for (uint32_t i = threadIdx.x; i < kCopies * 1024; i += kEntThreads) (&hc[0][0])[i] = 0;
    __syncthreads();
    // A workgroup takes every gridDim.x-th tile: what it adds to the image's counts in the end are atomics of all
    // workgroups on the same few addresses, and with one workgroup per tile those took 30 us for a 4K picture.
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t b = tile * kTileBlocks + threadIdx.x;
    const bool has_block = threadIdx.x < kTileBlocks && b < nbc;
    BlockRegs blk;
    const int ch = (int)(b % 3u);
    int prev_dc = 0;
    if (has_block) {
        if (kPlanar) {
            const uint32_t n_mcu = nbc / 3u, mcu = b / 3u;
            const uint4* q = reinterpret_cast<const uint4*>(coefs) + ((size_t)img * 24 + ch) * n_mcu + mcu;
            load_block_planar(blk, q, (size_t)3 * n_mcu, (uint32_t)chunk_masks[(size_t)img * nbc + b]);
            prev_dc = (b >= 3) ? (int)*reinterpret_cast<const int16_t*>(q - 1) : 0;
        } else {
            const int16_t* p = coefs + ((size_t)img * nbc + b) * 64;
            load_block(blk, p, chunk_masks != nullptr ? (uint32_t)chunk_masks[(size_t)img * nbc + b] : 0xffu);
            prev_dc = (b >= 3) ? (int)p[-192] : 0;
        }
    }
    if (has_block) {
        HistSink sink;
        sink.hist = h + (mode.is_lum((uint32_t)ch) ? 0 : 512);
        // the code tables only shape the bits, which this sink throws away: any readable words will do
        (void)encode_block(blk, prev_dc, h, h, sink);
    }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 1024; i += kEntThreads) {
        uint32_t n = 0;
#pragma unroll
        for (uint32_t c = 0; c < kCopies; ++c) n += hc[c][i];
        if (n != 0) atomicAdd(&hist[(size_t)img * 1024 + i], n);
    }
}

// id of an AC symbol in the ascending list of the 162 symbols the format has (sizes 1..10 with runs 0..15, the
// end-of-block symbol 0x00 and the run symbol 0xF0); -1 for the others.
__device__ __forceinline__ int ac_symbol_id(uint32_t s) {
    const uint32_t run = s >> 4, ss = s & 15u;
    if (s == 0x00u) return 0;
    if (s == 0xF0u) return 151;
    if (ss < 1u || ss > 10u) return -1;
    return (int)(1u + run * 10u + (ss - 1u) + (run == 15u ? 1u : 0u));
}

// One workgroup per table: blockIdx.x = table in header order (lum DC, lum AC, chr DC, chr AC), blockIdx.y = image.
// The sorts and the code assignment use all threads; the tree is one thread's work.
__global__ __launch_bounds__(kEntThreads) void build_huffman_kernel(const uint32_t* __restrict__ hist, HuffTables* __restrict__ tables,
                                                           uint8_t* __restrict__ spec) {
    constexpr int kMax = 163;
    __shared__ unsigned long long key[kMax + 1];
    __shared__ unsigned long long weight[2 * kMax];
    __shared__ uint16_t parent[2 * kMax];
    __shared__ uint16_t order[kMax + 1];
    __shared__ uint8_t syms[kMax];
    __shared__ uint32_t count[2 * kMax + 2];
    __shared__ uint16_t leaf_depth[kMax + 1];
    __shared__ uint8_t vals[kMax];
    constexpr uint32_t kT = kEntThreads;
    __shared__ int max_depth;
    const uint32_t t = blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
    const bool is_ac = (t & 1u) != 0;
    const uint32_t cls = t >> 1;
    const int n = is_ac ? 162 : 12, N = n + 1;
    const uint32_t* freq = hist + ((size_t)img * 4 + t) * 256;
    // 1. leaves: weight = occurrences + 1, and the extra leaf of weight 0; key ascending = (weight, id descending)
    for (uint32_t s = lane; s < 256; s += kT) {
        const int id = is_ac ? ac_symbol_id(s) : (s < 12u ? (int)s : -1);
        if (id >= 0) {
            syms[id] = (uint8_t)s;
            key[id] = ((unsigned long long)freq[s] + 1ull) << 9 | (unsigned long long)(511 - id);
        }
    }
    if (lane == 0) {
        key[n] = (unsigned long long)(511 - n);
        max_depth = 0;
    }
    for (uint32_t i = lane; i < 2 * kMax + 2; i += kT) count[i] = 0;
    __syncthreads();
    // 2. sort by counting smaller keys (the keys are distinct)
    for (int i = (int)lane; i < N; i += (int)kT) {
        const unsigned long long k = key[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) rank += key[j] < k ? 1 : 0;
        order[rank] = (uint16_t)i;
        weight[rank] = k >> 9;
    }
    __syncthreads();
    // two queues: leaves in sorted order (nodes 0..N-1), merged nodes in creation order (N..2N-2); the lighter head
    // is taken, the leaf on a tie
    if (lane == 0) {
        // the heads of both queues and the entries behind them stay in registers: the loop is one lane's chain of
        // dependent steps, and a read of LDS in it costs as much as twenty instructions
        constexpr unsigned long long kNone = ~0ull;
        int lh = 0, ih = N, made = N;
        unsigned long long l0 = weight[0], l1 = N > 1 ? weight[1] : kNone, i0 = kNone, i1 = kNone;
        for (int m = 0; m < N - 1; ++m) {
            int pick[2];
            unsigned long long sum = 0;
            for (int k = 0; k < 2; ++k) {
                if (l0 != kNone && (i0 == kNone || l0 <= i0)) {
                    pick[k] = lh++;
                    sum += l0;
                    l0 = l1;
                    l1 = lh + 1 < N ? weight[lh + 1] : kNone;
                } else {
                    pick[k] = ih++;
                    sum += i0;
                    i0 = i1;
                    i1 = ih + 1 < made ? weight[ih + 1] : kNone;
                }
            }
            weight[made] = sum;
            parent[pick[0]] = parent[pick[1]] = (uint16_t)made;
            // the new node joins the queue of merged nodes: it is its head, the one behind the head, or further back
            if (i0 == kNone) i0 = sum;
            else if (i1 == kNone && ih + 1 == made) i1 = sum;
            ++made;
        }
    }
    __syncthreads();
    // 3. depth of every leaf = steps up to the root, node 2N-2
    for (int i = (int)lane; i < N; i += (int)kT) {
        int v = i, d = 0;
        while (v != 2 * N - 2) {
            v = parent[v];
            ++d;
        }
        leaf_depth[order[i]] = (uint16_t)d;
        atomicAdd(&count[d], 1u);
        atomicMax(&max_depth, d);
    }
    __syncthreads();
    if (lane == 0) {
        // T.81 Figure K.3, with 11 bits for DC codes: a block's bit string then never exceeds the 1660 bits that
        // the Annex K codes can reach, which the slots are sized for
        const int limit = is_ac ? 16 : 11;
        for (int i = max_depth; i > limit; --i) {
            while (count[i] > 0) {
                int j = i - 2;
                while (count[j] == 0) --j;
                count[i] -= 2;
                count[i - 1] += 1;
                count[j + 1] += 2;
                count[j] -= 1;
            }
        }
        int i = limit;
        while (count[i] == 0) --i;
        count[i]--;                             // the extra leaf
    }
    // 4. symbols by (depth, id): the value list
    for (int i = (int)lane; i < n; i += (int)kT) {
        const uint32_t k = (uint32_t)leaf_depth[i] << 8 | (uint32_t)i;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += ((uint32_t)leaf_depth[j] << 8 | (uint32_t)j) < k ? 1 : 0;
        vals[rank] = syms[i];
    }
    __syncthreads();
    // the table as the header carries it, and the code words (T.81 Annex C)
    const uint32_t at = (t == 0 ? 0u : (t == 1 ? 29u : (t == 2 ? 208u : 237u)));
    uint8_t* out = spec + (size_t)img * kHuffSpecBytes + at;
    uint32_t* codes = is_ac ? tables[img].ac[cls] : tables[img].dc[cls];
    if (is_ac)
        for (uint32_t s = lane; s < 256; s += kT) codes[s] = 0;
    __syncthreads();
    if (lane == 0) out[0] = (uint8_t)((is_ac ? 0x10u : 0x00u) | cls);
    if (lane < 16) out[1 + lane] = (uint8_t)count[1 + lane];
    for (int i = (int)lane; i < n; i += (int)kT) out[17 + i] = vals[i];
    // T.81 Annex C: the k-th value gets the k-th code; every lane works out where its values fall
    for (int k = (int)lane; k < n; k += (int)kT) {
        uint32_t code = 0, first = 0;
        for (uint32_t l = 1; l <= 16; ++l) {
            const uint32_t c = count[l];
            if ((uint32_t)k < first + c) {
                codes[vals[k]] = (code + ((uint32_t)k - first)) | (l << 16);
                break;
            }
            code = (code + c) << 1;
            first += c;
        }
    }
}

// The tables' bytes into the header of every file (the files have been written with the standard header).
__global__ void patch_huffman_spec_kernel(uint8_t* __restrict__ out, const uint64_t* __restrict__ offsets,
                                          const uint8_t* __restrict__ spec, const int32_t* __restrict__ status) {
    if (*status != 0) return;
    uint8_t* file = out + offsets[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < kHuffSpecBytes; i += blockDim.x)
        file[kHuffSpecOffset + i] = spec[(size_t)blockIdx.x * kHuffSpecBytes + i];
}

}  // namespace

void launch_bit_lengths(hipStream_t s, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                        const HuffTables* d_huff, uint16_t* d_len16, uint32_t* d_tile_bits, int32_t* d_status,
                        ComponentMode mode) {
    const uint32_t nbc = (uint32_t)(mcus_per_image * 3);
    const uint32_t n_tiles = tiles_per_image(mcus_per_image);
    block_bits_kernel<false><<<dim3(n_tiles, (unsigned)n_images), kEntThreads, 0, s>>>(
        d_coefs, nbc, n_tiles, d_huff, d_len16, d_tile_bits, nullptr, d_status, StripeInfo{}, nullptr, mode);
}

void launch_entropy_encode(hipStream_t s, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                           const HuffTables* d_huff, const EntropyScratch& sc, const HeaderBytes& header,
                           int header_len, bool with_eoi, uint8_t* d_out, size_t out_capacity, uint64_t* d_offsets,
                           uint64_t* d_sizes, int32_t* d_status, const uint64_t* d_prev_offset,
                           const uint64_t* d_prev_size) {
    const uint32_t nbc = (uint32_t)(mcus_per_image * 3);
    const uint32_t n_tiles = tiles_per_image(mcus_per_image);
    const dim3 grid(n_tiles, (unsigned)n_images);
    const uint32_t n = (uint32_t)n_images;
    const uint32_t eoi = with_eoi ? 1u : 0u;

    const bool optimal = sc.optimal.tables != nullptr;
    if (optimal) {
        // two passes over the coefficients, as the reference's note says: count the symbols, make the tables
        (void)hipMemsetAsync(sc.optimal.hist, 0, n_images * 1024 * sizeof(uint32_t), s);
        uint32_t per_image = (256u + n - 1) / n;      // about one workgroup per compute unit over the whole job
        if (per_image > n_tiles) per_image = n_tiles;
        const dim3 hgrid(per_image, n);
        if (sc.planar)
            symbol_histogram_kernel<true><<<hgrid, kEntThreads, 0, s>>>(d_coefs, nbc, n_tiles, sc.chunk_masks, sc.mode, sc.optimal.hist);
        else
            symbol_histogram_kernel<false><<<hgrid, kEntThreads, 0, s>>>(d_coefs, nbc, n_tiles, sc.chunk_masks, sc.mode, sc.optimal.hist);
        build_huffman_kernel<<<dim3(4, n), kEntThreads, 0, s>>>(sc.optimal.hist, sc.optimal.tables, sc.optimal.spec);
        d_huff = sc.optimal.tables;
    }
    // The bit-length kernel also tallies the 0xFF bytes; one kernel packs and stuffs.
    if (sc.planar)
        tile_bits_kernel<true><<<grid, kEntThreads, 0, s>>>(d_coefs, nbc, n_tiles, d_huff, sc.chunk_masks, sc.mode,
                                                            sc.len16, sc.slots, sc.tile_bits, (uint4*)sc.tile_ff8,
                                                            d_status, optimal ? 1u : 0u);
    else
        tile_bits_kernel<false><<<grid, kEntThreads, 0, s>>>(d_coefs, nbc, n_tiles, d_huff, sc.chunk_masks, sc.mode,
                                                             sc.len16, sc.slots, sc.tile_bits, (uint4*)sc.tile_ff8,
                                                             d_status, optimal ? 1u : 0u);
    if (sc.marks != nullptr) (void)hipEventRecord(sc.marks[0], s);
    if (n == 1 && n_tiles <= kFoldTiles && d_prev_offset == nullptr) {
        // One image with few tiles: every workgroup sums over the tiles before it, which is cheaper than a
        // single-workgroup scan launched in between.
        pack_stuff_kernel<true><<<grid, kEntThreads, 0, s>>>(nbc, n_tiles, sc.slots, sc.len16, sc.tile_bits,
                                                             (const uint4*)sc.tile_ff8, nullptr, nullptr, d_offsets,
                                                             d_sizes, header, (uint32_t)header_len, eoi, d_out,
                                                             out_capacity, d_status);
        if (optimal && header_len) patch_huffman_spec_kernel<<<n, 256, 0, s>>>(d_out, d_offsets, sc.optimal.spec, d_status);
        return;
    }
    ImageLevel lvl{2, sc.img_bits, d_offsets, d_sizes, (uint64_t)out_capacity, (uint32_t)header_len + 2 * eoi, d_status};
    lvl.prev_offset = d_prev_offset;
    lvl.prev_size = d_prev_size;
    if (sc.wait_before_offsets != nullptr) (void)hipStreamWaitEvent(s, sc.wait_before_offsets, 0);
    scan_tiles_ff_kernel<<<n, 1024, 0, s>>>(sc.tile_bits, (const uint4*)sc.tile_ff8, n_tiles, sc.tile_base,
                                            sc.tile_ffbase, sc.img_bits, sc.img_ff, lvl);
    if (n > 1) scan_images_kernel<<<1, 1024, 0, s>>>(n, sc.img_ff, lvl);
    if (sc.record_after_offsets != nullptr) (void)hipEventRecord(sc.record_after_offsets, s);
    pack_stuff_kernel<false><<<grid, kEntThreads, 0, s>>>(nbc, n_tiles, sc.slots, sc.len16, sc.tile_bits,
                                                          (const uint4*)sc.tile_ff8, sc.tile_base, sc.tile_ffbase,
                                                          d_offsets, d_sizes, header, (uint32_t)header_len, eoi, d_out,
                                                          out_capacity, d_status);
    if (optimal && header_len) patch_huffman_spec_kernel<<<n, 256, 0, s>>>(d_out, d_offsets, sc.optimal.spec, d_status);
}

namespace {
// First 32 bits (MSB first, zero padded) of the stream that the blocks' strings form.
// d_facts (may be null): also leaves {the stripe's bit count, its head bits} there as two 64-bit words -- what the
// stripes' owners exchange before the last step.
__global__ void stripe_head_kernel(const uint16_t* __restrict__ len16, const uint32_t* __restrict__ slots,
                                   uint32_t nbc, uint32_t* __restrict__ head, const uint64_t* __restrict__ img_bits,
                                   uint64_t* __restrict__ d_facts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t acc = 0, n = 0;
    for (uint32_t b = 0; b < nbc && n < 32; ++b) {
        const uint32_t len = len16[b];
        const uint32_t* slot = slots + (size_t)(b / kTileBlocks) * kSlotWords * kSlotStride + (b % kTileBlocks);
        for (uint32_t i = 0; i < len && n < 32; ++i, ++n) {
            const uint32_t word = slot[(size_t)(i >> 5) * kSlotStride];
            acc |= ((word >> (31u - (i & 31u))) & 1u) << (31u - n);
        }
    }
    *head = acc;
    if (d_facts != nullptr) {
        d_facts[0] = img_bits[0];
        d_facts[1] = acc;
    }
}

// The last MCU's quantised DC values (Y, Cb, Cr; a fourth word of padding) for the owner of the stripe below.
__global__ void stripe_last_dc_kernel(const int16_t* __restrict__ coefs, uint32_t mcus, int16_t* __restrict__ d_last_dc) {
    if (threadIdx.x < 4) d_last_dc[threadIdx.x] = threadIdx.x < 3 ? coefs[((size_t)(mcus - 1) * 3 + threadIdx.x) * 64] : (int16_t)0;
}

// Where stripe `index` of `n` lies in the image's stream, from every stripe's {bit count, head bits}: the sum of the
// bit counts above it (bit_stream.cpp:44-67 appends, so a stripe starts where the one before ends) and the head bits
// of the stripe below.  One wave; the stripes are few.
__global__ void stripe_place_kernel(const uint64_t* __restrict__ facts, uint32_t index, uint32_t n, StripeInfo* __restrict__ st) {
    uint64_t before = 0;
    for (uint32_t r = threadIdx.x; r < index; r += 64) before += facts[2 * (size_t)r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if (threadIdx.x != 0) return;
    StripeInfo s{};
    s.bit_base = before;
    s.first_byte = (before + 7) >> 3;
    s.origin_byte = s.first_byte & ~uint64_t(3);
    s.has_next = index + 1 < n ? 1u : 0u;
    s.next_head = s.has_next ? (uint32_t)facts[2 * (size_t)(index + 1) + 1] : 0u;
    s.active = 1;
    *st = s;
}
}  // namespace

void launch_stripe_last_dc(hipStream_t s, const int16_t* d_coefs, size_t mcus, int16_t* d_last_dc) {
    stripe_last_dc_kernel<<<1, 64, 0, s>>>(d_coefs, (uint32_t)mcus, d_last_dc);
}

void launch_stripe_bits(hipStream_t s, const int16_t* d_coefs, size_t mcus, const HuffTables* d_huff,
                        const EntropyScratch& sc, const StripeInfo& st, uint32_t* d_head, int32_t* d_status,
                        const int16_t* d_initial_dc, uint64_t* d_facts) {
    const uint32_t nbc = (uint32_t)(mcus * 3);
    const uint32_t n_tiles = tiles_per_image(mcus);
    block_bits_kernel<true><<<dim3(n_tiles, 1), kEntThreads, 0, s>>>(d_coefs, nbc, n_tiles, d_huff, sc.len16, sc.tile_bits,
                                                                     sc.slots, d_status, st, sc.chunk_masks, sc.mode,
                                                                     d_initial_dc);
    ImageLevel l1{1, nullptr, sc.img_ubase, nullptr, (uint64_t)sc.unstuffed_capacity, 0, d_status};
    scan_tiles_kernel<<<1, 1024, 0, s>>>(sc.tile_bits, n_tiles, sc.tile_base, sc.img_bits, l1);
    stripe_head_kernel<<<1, 64, 0, s>>>(sc.len16, sc.slots, nbc, d_head, sc.img_bits, d_facts);
}

void launch_stripe_place(hipStream_t s, const uint64_t* d_all_facts, uint32_t index, uint32_t n_stripes, StripeInfo* d_st) {
    stripe_place_kernel<<<1, 64, 0, s>>>(d_all_facts, index, n_stripes, d_st);
}

void launch_stripe_finish(hipStream_t s, size_t mcus, const EntropyScratch& sc, const StripeInfo& st,
                          const HeaderBytes& header, int header_len, bool with_eoi, uint8_t* d_out,
                          size_t out_capacity, uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status,
                          const StripeInfo* d_st) {
    const uint32_t nbc = (uint32_t)(mcus * 3);
    const uint32_t n_tiles = tiles_per_image(mcus);
    const dim3 grid(n_tiles, 1);
    const uint32_t eoi = with_eoi ? 1u : 0u;
    pack_kernel<<<grid, kEntThreads, 0, s>>>(nbc, n_tiles, sc.slots, sc.len16, sc.tile_bits, sc.tile_base,
                                                    sc.img_bits, sc.img_ubase, sc.unstuffed, sc.unstuffed_capacity,
                                                    sc.tile_ff, d_status, st, d_st);
    // the scan's own size estimate is only a capacity check here; the stuff pass writes the exact segment size
    ImageLevel l2{2, sc.img_bits, d_offsets, d_sizes, (uint64_t)out_capacity + 8, (uint32_t)header_len + 2 * eoi, d_status};
    scan_tiles_kernel<<<1, 1024, 0, s>>>(sc.tile_ff, n_tiles, sc.tile_ffbase, sc.img_ff, l2);
    stuff_kernel<<<grid, kEntThreads, 0, s>>>(n_tiles, sc.tile_bits, sc.tile_base, sc.img_bits, sc.img_ubase,
                                                     sc.unstuffed, sc.tile_ff, sc.tile_ffbase, sc.img_ff, d_offsets,
                                                     d_sizes, header, (uint32_t)header_len, eoi, d_out, out_capacity,
                                                     d_status, st, d_st);
}

}  // namespace jea
