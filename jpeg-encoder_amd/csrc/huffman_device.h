// Device functions of the Huffman coder shared by the image-level kernels
// (entropy.hip) and the per-block kernels (blocks.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/jpegenc_amd.h"
#include "kernels.h"

namespace jea {

// A barrier for threads that have exchanged data through LDS only.  __syncthreads() also orders global memory:
// it waits for every load and store the wave has under way, so a loop that meets a barrier pays the latency of its
// memory operations once per turn, and nothing asked for ahead of time stays ahead.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr uint32_t kStreamWin = 2048;  // bytes of compressed stream staged in LDS by the serial decoder

struct HuffLds {
    uint32_t dc[2][12];
    uint32_t ac[2][256];
};

// kGroup: threads of the workgroup.
template <uint32_t kGroup>
__device__ __forceinline__ void load_huff(HuffLds& h, const HuffTables* __restrict__ g) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&h);
    // all loads are issued before the first store (as one loop each piece would wait for the one before it)
    constexpr uint32_t kWords = sizeof(HuffLds) / 4, kPerThread = (kWords + kGroup - 1) / kGroup;
    uint32_t v[kPerThread];
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) {
        const uint32_t i = r * kGroup + threadIdx.x;
        if (i < kWords) v[r] = src[i];
    }
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) {
        const uint32_t i = r * kGroup + threadIdx.x;
        if (i < kWords) dst[i] = v[r];
    }
}

// The 64 zig-zag coefficients of one block-channel as loaded (8 x 16 bytes).
struct BlockRegs {
    uint32_t w[32];
    __device__ __forceinline__ int at(int k) const { return (int)(int16_t)(w[k >> 1] >> (16 * (k & 1))); }
};

// `chunk_mask` bit i clear = the i-th 16-byte chunk is known to be all zero and is not read.
__device__ __forceinline__ void load_block(BlockRegs& r, const int16_t* __restrict__ p, uint32_t chunk_mask = 0xffu) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((chunk_mask >> i) & 1u) v = q[i];
        r.w[4 * i + 0] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
    }
}

// Chunk-planar layout: chunk i of the block is q[i * plane_stride] (16-byte units).
__device__ __forceinline__ void load_block_planar(BlockRegs& r, const uint4* __restrict__ q, size_t plane_stride,
                                                  uint32_t chunk_mask) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i == 0 || ((chunk_mask >> i) & 1u)) v = q[(size_t)i * plane_stride];   // chunk 0 is always stored
        r.w[4 * i + 0] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
    }
}

__device__ __forceinline__ uint32_t bit_width(uint32_t v) { return 32u - (uint32_t)__clz((int)v); }

// Symbol generation for one block-channel, shared by the length pass and the
// pack pass so that both see the same symbols.  Sink::put(bits, n) receives the
// n low bits of `bits`, n in 1..27.
//   DC  entropy_encoder.cpp:97, 715-730: SSSS = bit_width(|diff|); code; then the
//       low SSSS bits of diff (diff > 0) or of ~|diff| (diff <= 0)
//   AC  :101-118: (run, v) on every non-zero v, (15, 0) for each full run of 16
//       zeros that is followed by a non-zero; EOB iff zz[63] == 0;  :732-765 codes
// Sink::symbol(is_ac, symbol) is told every symbol before its bits (the histogram pass of the
// per-image tables listens; the other sinks ignore it).
// Returns false when a value falls outside the tables (SSSS > 11 / > 10).
template <class Sink>
__device__ __forceinline__ bool encode_block(const BlockRegs& blk, int prev_dc, const uint32_t* __restrict__ dc_tab,
                                             const uint32_t* __restrict__ ac_tab, Sink& sink) {
    bool ok = true;
    {
        const int diff = blk.at(0) - prev_dc;
        const uint32_t amp = (uint32_t)(diff < 0 ? -diff : diff);
        uint32_t s = bit_width(amp);
        if (s > 11) { ok = false; s = 11; }
        const uint32_t e = dc_tab[s];
        const uint32_t extra = (diff > 0 ? amp : ~amp) & ((1u << s) - 1u);
        sink.symbol(false, s);
        sink.put(((e & 0xffffu) << s) | extra, (e >> 16) + s);
    }
    uint32_t run = 0;
    uint32_t largest = 0;   // largest AC magnitude: the range is checked once, after the walk
    // Coefficients are walked in groups of eight (one 16-byte load); a group that is
    // all zero only extends the run.  High zig-zag positions are zero in most
    // blocks, and a wave skips the group body when none of its lanes needs it.
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const uint32_t any = (g == 0 ? (blk.w[0] & 0xffff0000u) : blk.w[4 * g]) | blk.w[4 * g + 1] | blk.w[4 * g + 2] |
                             blk.w[4 * g + 3];
        if (any == 0) {
            run += (g == 0) ? 7u : 8u;
            continue;
        }
#pragma unroll
        for (int k = (g == 0 ? 1 : 8 * g); k < 8 * g + 8; ++k) {
            const int v = blk.at(k);
            if (v != 0) {
                while (run >= 16) {
                    const uint32_t z = ac_tab[0xF0];
                    sink.symbol(true, 0xF0u);
                    sink.put(z & 0xffffu, z >> 16);
                    run -= 16;
                }
                const uint32_t amp = (uint32_t)(v < 0 ? -v : v);
                largest = amp > largest ? amp : largest;
                uint32_t s = bit_width(amp);
                s = s < 10u ? s : 10u;           // keeps the table index and the bit count in range when the value is not
                const uint32_t e = ac_tab[(run << 4) | s];
                // the low s bits of v (v > 0) or of v - 1 (v < 0), entropy_encoder.cpp:745-765
                const uint32_t extra = __builtin_amdgcn_ubfe((uint32_t)(v + (v >> 31)), 0u, s);
                sink.symbol(true, (run << 4) | s);
                sink.put(((e & 0xffffu) << s) | extra, (e >> 16) + s);
                run = 0;
            } else {
                ++run;
            }
        }
    }
    if (largest > 1023u) ok = false;
    if (run > 0) {  // zz[63] == 0: trailing zeros (and their ZRLs) collapse into EOB
        const uint32_t e = ac_tab[0x00];
        sink.symbol(true, 0x00u);
        sink.put(e & 0xffffu, e >> 16);
    }
    return ok;
}

struct CountSink {
    uint32_t bits = 0;
    __device__ __forceinline__ void put(uint32_t, uint32_t n) { bits += n; }
    __device__ __forceinline__ void symbol(bool, uint32_t) {}
    __device__ __forceinline__ void finish() {}
};

// Counts symbols: hist[symbol] for DC symbols, hist[256 + symbol] for AC symbols (LDS, shared by the workgroup).
struct HistSink {
    uint32_t* hist;
    __device__ __forceinline__ void put(uint32_t, uint32_t) {}
    __device__ __forceinline__ void symbol(bool is_ac, uint32_t sym) { atomicAdd(&hist[(is_ac ? 256u : 0u) + sym], 1u); }
    __device__ __forceinline__ void finish() {}
};

// MSB-first writer into the block's slot: full words go out as they complete,
// the last partial word is zero padded.
struct SlotSink {
    uint32_t* slot;  // first word of the block's string; consecutive words are `stride` apart
    uint32_t stride = kSlotStride;
    uint32_t bits = 0;
    uint32_t acc = 0;    // the word being filled, MSB first
    uint32_t nacc = 0;   // bits of it in use, 0..31
    // 32-bit arithmetic throughout: a 64-bit accumulator costs a 64-bit shift per symbol, which issues at a quarter
    // of the rate of the 32-bit ones
    __device__ __forceinline__ void put(uint32_t v, uint32_t n) {
        const uint32_t t = v << (32u - n);       // the n bits at the top of a word (n is 1..27)
        acc |= t >> nacc;
        const uint32_t used = nacc;
        nacc += n;
        bits += n;
        if (nacc >= 32) {
            *slot = acc;
            slot += stride;
            acc = __builtin_amdgcn_alignbit(t, 0u, used);   // what did not fit: t << (32 - used); nothing when used == 0
            nacc -= 32;
        }
    }
    __device__ __forceinline__ void symbol(bool, uint32_t) {}
    __device__ __forceinline__ void finish() {
        if (nacc > 0) *slot = acc;
    }
};

// ---------------------------------------------------------------------------
// The same symbols, for the kernel that writes every block's string (tile_bits_kernel), with the work per
// coefficient cut down to what the vector ALU cannot be spared: that kernel is bound by instruction issue,
// and two thirds of its instructions are this loop's body.
//   * 8-byte table entries: .x = the code word at the top of a word with its length (at most 16) in the low five
//     bits, .y = that length plus the number of amplitude bits that follow the code (which the symbol names: its
//     low four bits, or the DC symbol itself).  One LDS read then gives the string's head in place, the shift
//     that puts the amplitude bits behind it (shifts take the low five bits of their operand) and the string's length.
//   * the size category of v from the leading bits of t = v - (v < 0), which the amplitude needs anyway:
//     for v < 0, ~t = |v|, so t's leading ones are |v|'s leading zeros (entropy_encoder.cpp:745-765 takes the
//     low bits of v or of v - 1).
//   * the run is kept as the table address of (run, 0).
// ---------------------------------------------------------------------------
// An AC row holds a run's symbols by falling size category: (run, s) is entry 16 run + 16 - s, so that a value whose
// t has f = 32 - s leading sign bits is found at 8 f bytes from a per-run address, one shift-and-add.  End-of-block
// (0, 0) and the run of sixteen zeros (15, 0) sit at entries 16 and 256: the places of the size 16 no symbol has.
struct HuffTop {
    static constexpr uint32_t kAcEntries = 257, kEob = 16, kZrl = 256;
    uint2 dc[2][12];
    uint2 ac[2][kAcEntries + 1];
};

template <uint32_t kGroup>
__device__ __forceinline__ void load_huff_top(HuffTop& h, const HuffTables* __restrict__ g) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    constexpr uint32_t kEntries = sizeof(HuffTables) / 4, kPerThread = (kEntries + kGroup - 1) / kGroup;
    uint32_t v[kPerThread];
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) {
        const uint32_t i = r * kGroup + threadIdx.x;
        if (i < kEntries) v[r] = src[i];
    }
#pragma unroll
    for (uint32_t r = 0; r < kPerThread; ++r) {
        const uint32_t i = r * kGroup + threadIdx.x;
        if (i < kEntries) {
            const uint32_t len = v[r] >> 16, code = v[r] & 0xffffu;
            const uint32_t j = i - 24u, symbol = j & 255u;    // AC entries
            const uint32_t amplitude_bits = i < 24u ? i % 12u : symbol & 15u;
            const uint2 e = make_uint2((len ? code << (32u - len) : 0u) | len, len + amplitude_bits);
            if (i < 24u) h.dc[i / 12u][i % 12u] = e;
            else h.ac[j >> 8][(symbol >> 4) * 16u + 16u - (symbol & 15u)] = e;
        }
    }
}

// MSB-first writer into the block's slot that takes its strings at the top of a word.  The slot's words are
// `kSlotStride` words apart; `base` is the same for the whole workgroup, so the stores take it from scalar registers
// and a 32-bit offset per lane.
struct TopSink {
    uint8_t* base;
    uint32_t offset;     // byte offset of the word being filled
    uint32_t offset0;
    uint32_t acc = 0;    // the word being filled
    uint32_t nacc = 0;   // bits of it in use, 0..31
    __device__ __forceinline__ TopSink(uint32_t* tile_slots, uint32_t column) : base(reinterpret_cast<uint8_t*>(tile_slots)), offset(4u * column), offset0(4u * column) {}
    // the n bits at the top of t, n in 1..27, zeros below them
    __device__ __forceinline__ void put_top(uint32_t t, uint32_t n) {
        acc |= t >> nacc;
        const uint32_t used = nacc;
        nacc += n;
        if (nacc >= 32) {
            *reinterpret_cast<uint32_t*>(base + offset) = acc;
            offset += 4u * kSlotStride;
            acc = __builtin_amdgcn_alignbit(t, 0u, used);   // what did not fit: t << (32 - used); nothing when used == 0
            nacc -= 32;
        }
    }
    __device__ __forceinline__ void finish() {
        if (nacc > 0) *reinterpret_cast<uint32_t*>(base + offset) = acc;
    }
    __device__ __forceinline__ uint32_t bits() const { return (offset - offset0) / (4u * kSlotStride) * 32u + nacc; }
};

// encode_block for TopSink and HuffTop; the same strings, bit for bit.  Returns false when a value falls outside
// the tables (its string is then some in-range value's).
// kChecked = false: the AC values are known to lie within +-1023 and are not looked at again.  That holds for what the
// library's own quantiser makes of 8-bit samples: an AC coefficient of the 8x8 transform of values in -128..127 is at
// most 1020 in magnitude before the division (128 times the sum of the magnitudes of the basis function's 64 values
// is 1024 for every one of them; half of the values meet 127 instead), and the divisors are at least 1.
template <bool kChecked>
__device__ __forceinline__ bool encode_block_top(const BlockRegs& blk, int prev_dc, const uint2* __restrict__ dc_tab,
                                                 const uint2* __restrict__ ac_tab, TopSink& sink) {
    bool ok = true;
    {
        const int diff = blk.at(0) - prev_dc;
        const uint32_t amp = (uint32_t)(diff < 0 ? -diff : diff);
        uint32_t s = bit_width(amp);
        if (s > 11) { ok = false; s = 11; }
        const uint2 e = dc_tab[s];
        const uint32_t extra = (diff > 0 ? amp : ~amp) & ((1u << s) - 1u);
        sink.put_top((e.x & ~31u) | __builtin_amdgcn_alignbit(extra, 0u, e.y), e.y);
    }
    // 32-bit LDS addresses, as numbers (a generic pointer is 64 bits wide, and so is every sum and comparison on it;
    // as pointers the compiler splits them into base and offset again and adds the two at every read).
    typedef __attribute__((address_space(3))) const uint8_t* LdsBytes;
    typedef uint32_t Pair __attribute__((ext_vector_type(2)));   // a built-in vector type: loads through an LDS pointer
    typedef __attribute__((address_space(3))) const Pair* LdsEntry;
    const uint32_t table = (uint32_t)(uintptr_t)(LdsBytes) reinterpret_cast<const uint8_t*>(ac_tab);
    // at_run + 8 f is the entry (run, 32 - f): entry 16 run + 16 - (32 - f)
    const uint32_t run_0 = table - 16u * 8u;
    const uint32_t zrl_from = run_0 + 16 * 128;   // from run 16 on
    uint32_t at_run = run_0;
    uint32_t fewest_sign_bits = 32;               // over the AC values: 32 - size category
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const uint32_t any = (g == 0 ? (blk.w[0] & 0xffff0000u) : blk.w[4 * g]) | blk.w[4 * g + 1] | blk.w[4 * g + 2] |
                             blk.w[4 * g + 3];
        if (any == 0) {
            at_run += (g == 0) ? 7 * 128 : 8 * 128;
            continue;
        }
#pragma unroll
        for (int k = (g == 0 ? 1 : 8 * g); k < 8 * g + 8; ++k) {
            const int v = blk.at(k);
            if (v != 0) {
                while (at_run >= zrl_from) {
                    const Pair z = *(LdsEntry)(uintptr_t)(table + HuffTop::kZrl * 8u);
                    sink.put_top(z.x & ~31u, z.y);
                    at_run -= 16 * 128;
                }
                const int t = v + (v >> 31);
                uint32_t f;   // leading bits of t that equal its sign (t is neither 0 nor -1)
                asm("v_ffbh_i32 %0, %1" : "=v"(f) : "v"(t));
                if (kChecked) {
                    fewest_sign_bits = f < fewest_sign_bits ? f : fewest_sign_bits;
                    f = f > 22u ? f : 22u;                           // keeps the entry and the bit count in range when the value is not
                }
                const Pair e = *(LdsEntry)(uintptr_t)(at_run + (f << 3));
                const uint32_t low_bits_on_top = (uint32_t)t << f;   // the 32 - f low bits of t
                sink.put_top((e.x & ~31u) | (low_bits_on_top >> e.x), e.y);
                at_run = run_0;
            } else {
                at_run += 128;
            }
        }
    }
    if (fewest_sign_bits < 22u) ok = false;
    if (at_run != run_0) {  // zz[63] == 0: trailing zeros (and their ZRLs) collapse into EOB
        const Pair e = *(LdsEntry)(uintptr_t)(table + HuffTop::kEob * 8u);
        sink.put_top(e.x & ~31u, e.y);
    }
    return ok;
}

// Places j of w0 (bit 31-j) from which on w0:w1 holds eight ones: where a 0xFF byte starts if a byte boundary falls
// there.  Bit 31-j of the result.
__device__ __forceinline__ uint32_t ff_starts(uint32_t w0, uint32_t w1) {
    // runs of 2, 4, 8 ones by doubling; of the low word only the top bits matter, which its own shifts get right
    const uint32_t a0 = w0 & __builtin_amdgcn_alignbit(w0, w1, 31), a1 = w1 & (w1 << 1);
    const uint32_t b0 = a0 & __builtin_amdgcn_alignbit(a0, a1, 30), b1 = a1 & (a1 << 2);
    return b0 & __builtin_amdgcn_alignbit(b0, b1, 28);
}

// Tallies of such places by j mod 8, over the words of one string.  A word's places are counted in 4-bit fields,
// all eight residues of all four bytes with four masks and four sums (a population count per residue is sixteen
// instructions and eight registers); a field holds up to fifteen words' worth and is then flushed into the
// per-residue totals, the four fields of a residue summed by one sum-of-bytes instruction.
struct FfTally {
    uint32_t field[4] = {0, 0, 0, 0};   // field[k], nibble n: places at bit 4 n + k
    uint32_t total[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // by j mod 8
    __device__ __forceinline__ void add(uint32_t m) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) field[k] += (m >> k) & 0x11111111u;
    }
    // after at most fifteen add()
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            // bit p is place j = 31 - p: even nibbles (p mod 8 = k) are residue 7 - k, odd ones residue 3 - k
            total[7 - k] = __builtin_amdgcn_sad_u8(field[k] & 0x0f0f0f0fu, 0u, total[7 - k]);
            total[3 - k] = __builtin_amdgcn_sad_u8((field[k] >> 4) & 0x0f0f0f0fu, 0u, total[3 - k]);
            field[k] = 0;
        }
    }
};

// Bit reader over a stream from which the stuffed zeros have already been removed (the per-block entry points:
// their callers unstuff first, as encoder.cpp:49 does).
struct SerialReader {
    const uint8_t* lds;     // staged bytes [base, base + kStreamWin + 16)
    uint64_t base;          // stream index of lds[0]
    uint64_t pos;           // next byte to take
    uint64_t len;           // stream length
    uint64_t buf;           // bit buffer, MSB first
    int nbits;              // valid bits in buf
    int phantom;            // zero bytes appended after the end of the stream
    __device__ __forceinline__ uint32_t raw(uint64_t i) const { return i < len ? lds[i - base] : 0u; }
    __device__ __forceinline__ void refill() {
        while (nbits <= 56) {
            uint32_t b = 0;
            if (pos < len) {
                b = raw(pos);
            } else {
                ++phantom;
            }
            ++pos;
            buf |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(buf >> (64 - n)); }
    __device__ __forceinline__ void skip(int n) { buf <<= n; nbits -= n; }
    // true when bits beyond the end of the file have been consumed
    __device__ __forceinline__ bool past_end() const { return nbits < 8 * phantom; }
};

// entropy_encoder.cpp:786-806, :845-864
__device__ __forceinline__ int read_amplitude(SerialReader& r, int ssss) {
    const uint32_t bits = r.peek(ssss);
    r.skip(ssss);
    const uint32_t mask = (1u << ssss) - 1u;
    if (bits >> (ssss - 1)) return (int)(bits & mask);
    return -(int)(mask & ~bits);
}

// The reference grows a candidate code bit by bit from two bits until it equals
// a code word of that length (entropy_encoder.cpp:767-778, :810-828); for these
// prefix codes that is the canonical decode, done here with two table reads.
// Returns (length << 8) | symbol of the code word at the top of the 16 bits, 0 when none matches.
__device__ __forceinline__ uint32_t huff_entry(const DecodeTable& t, uint32_t p16) {
    uint32_t e = t.fast[p16 >> 7];
    if (e == 0) {
        const uint32_t idx = p16 - t.first_long;   // wraps to a huge value below first_long
        e = idx < (uint32_t)kDecodeSlowEntries ? t.slow[idx] : 0u;
    }
    return e;
}

// Returns the symbol, -1 on an invalid code.
__device__ __forceinline__ int decode_symbol(SerialReader& r, const DecodeTable& t) {
    const uint32_t e = huff_entry(t, r.peek(16));
    if (e == 0) return -1;
    r.skip((int)(e >> 8));
    return (int)(e & 0xffu);
}

// One block-channel from the serial reader into zz[64] (zig-zag order; the caller
// has zeroed it): symbol loop of entropy_encoder.cpp:703-712, run-length removal
// of :122-144 -- which restores every entry except the LAST one, so a block that
// ends without EOB loses its final coefficient -- and the DC chain (:125-126).
// Returns 0 or JEA_ERR_BAD_STREAM.
__device__ __forceinline__ int decode_block(SerialReader& rd, const DecodeTable& dct, const DecodeTable& act,
                                            int& last_dc, int16_t* zz) {
    int err = 0;
    rd.refill();
    const int s = decode_symbol(rd, dct);
    int diff = 0;
    if (s < 0 || s > 11) err = JEA_ERR_BAD_STREAM;
    else if (s > 0) { rd.refill(); diff = read_amplitude(rd, s); }
    int k = 1, processed = 0;
    bool pending = false;
    int pend_run = 0, pend_val = 0;
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
        if (pending) {  // the previous entry is now known not to be the last: restore it
            k += pend_run;
            if (k < 64) zz[k] = (int16_t)pend_val;
            ++k;
        }
        pending = true; pend_run = run; pend_val = val;
        processed += 1 + run;
        if (processed > 63) { err = JEA_ERR_BAD_STREAM; break; }
        if ((run == 0 && val == 0) || processed == 63) break;
    }
    if (rd.past_end()) err = JEA_ERR_BAD_STREAM;
    last_dc = (int)(int16_t)(diff + last_dc);
    zz[0] = (int16_t)last_dc;
    return err;
}

}  // namespace jea
