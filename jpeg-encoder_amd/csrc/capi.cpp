// C-ABI of the library (include/jpegenc_amd.h): argument checks, scratch
// management and the order in which the kernels are enqueued.  All arithmetic on
// image data happens in the kernels (fdct.hip, entropy.hip); the only host-side
// computation is table and header construction, which the reference also does
// once per encoder object / per file on the host
// (jpeg/src/quantiser.cpp:3-55, jpeg/src/encoder.cpp:74-138,140-283).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jpegenc_amd.h"
#include "kernels.h"
#include "tables.h"

using namespace jea;

// Scratch of jea_encode_batch_static_async: device memory that exists from the moment the library is loaded,
// one copy per GPU, so that the call itself allocates nothing.  In host code this name is only a handle for
// hipGetSymbolAddress; the address of a GPU's copy is looked up once per device (static_scratch_of) and
// only that address is ever handed to kernels.
constexpr size_t kStaticScratchBytes = (size_t)JEA_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

thread_local std::string g_create_error;

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
// Recorded behind every call that uses a GPU's copy and waited for by the next one, whichever context or
// stream that comes from: users of the one buffer run one after the other.  Created with the address, kept
// until the process ends.
hipEvent_t g_scratch_free[kMaxDevices] = {};

// Address of `scratch` on the current device (which must be `device`); asked for once per device.
hipError_t static_scratch_of(int device, unsigned char** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&g_scratch_free[device], hipEventDisableTiming)) != hipSuccess) return e;
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace

// One of the two staging slots of the host-to-host batch entry points: while the kernels work on one slot's piece, the
// next piece travels to the other slot and the piece before travels back.
struct PipeSlot {
    DeviceBuffer d_in, d_out, d_meta;
    void* pin_in = nullptr;
    size_t pin_in_bytes = 0;
    void* pin_out = nullptr;
    size_t pin_out_bytes = 0;
    void* pin_meta = nullptr;
    size_t pin_meta_bytes = 0;
    hipEvent_t h2d = nullptr, work = nullptr, d2h = nullptr;
};

struct jea_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;
    HuffTables* d_huff = nullptr;
    DecodeTables* d_decode = nullptr;
    StepTables* d_steps = nullptr;
    int32_t* d_status = nullptr;
    ComponentMode mode;   // jea_set_colour_mapper
    bool optimal_huffman = false;   // jea_set_huffman_optimised
    unsigned char* static_scratch = nullptr;   // this GPU's copy of `scratch`
    // jea_set_profiling: events recorded at the phase boundaries of the last encode / decode call
    size_t scratch_budget = (size_t)32 << 30;   // jea_set_scratch_budget
    bool profiling = false;
    hipEvent_t marks[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n_marks = 0;   // marks recorded by the last profiled call
    // scratch, grown on demand and kept
    DeviceBuffer coefs, masks, len16, slots, tile_bits, tile_base, tile_ff, tile_ffbase, img_words, unstuffed;
    DeviceBuffer stage_in, stage_out, seg_offsets, out_offsets, out_sizes;
    DeviceBuffer stream_copy, small;  // jea_stream_upload / per-block entry points
    // parallel decode
    DeviceBuffer ustream, chunk_drops, chunk_base, img_drops, sub_end, sub_used, sub_info, grp_sum, grp_base,
        round_flags, grp_pending, grp_end, sub_guess, first_bad, ctx_keys, ctx_vals, huff_hist, huff_tables, huff_spec,
        step_tables;
    DeviceBuffer tile_ff8, group_flags;
    void* pin_io = nullptr;   // pinned staging for the host-buffer entry points
    size_t pin_io_bytes = 0;
    hipEvent_t piece_done[4] = {nullptr, nullptr, nullptr, nullptr};
    // jea_encode_batch_async on large batches: a second stream and the events that tie it to `stream`
    hipStream_t lane_stream = nullptr;
    hipEvent_t lane_fork = nullptr, lane_join = nullptr, lane_offsets[2] = {nullptr, nullptr};
    // jea_encode_host_batch / jea_decode_host_batch: copy streams and staging
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    PipeSlot pipe[2];
    void* pinned = nullptr;   // host memory the GPU reads and writes: decode bookkeeping
    size_t pinned_bytes = 0;
    size_t stream_len = 0;
    // stripe encode in progress (jea_stripe_*)
    size_t stripe_mcus = 0;
    uint16_t stripe_height = 0;
    int stripe_stage = 0;  // 0 none, 1 transformed, 2 bits known
    size_t stripe_capacity = 0;
};

namespace {

int fail(jea_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return code;
}

#define JEA_HIP(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, JEA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

int ensure(jea_ctx* ctx, DeviceBuffer& b, size_t bytes) {
    if (bytes <= b.bytes) return JEA_OK;
    // the context's GPU must be current: hipMalloc takes whichever device is (several contexts on several
    // GPUs may be driven from one thread)
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    // work that still uses the old allocation must drain before it is freed
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (b.ptr) JEA_HIP(ctx, hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.ptr, want);
    if (e == hipErrorOutOfMemory) {   // without the head-room
        (void)hipGetLastError();
        want = bytes;
        e = hipMalloc(&b.ptr, want);
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        b.ptr = nullptr;
        return fail(ctx, JEA_ERR_CAPACITY, "scratch buffer of " + std::to_string(bytes) +
                                               " bytes does not fit the GPU's free memory; encode or decode fewer "
                                               "images per call, or lower jea_set_scratch_budget");
    }
    JEA_HIP(ctx, e);
    b.bytes = want;
    return JEA_OK;
}

int ensure_pinned(jea_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return JEA_OK;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pinned) JEA_HIP(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    JEA_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
    return JEA_OK;
}

int ensure_pin_io(jea_ctx* ctx, size_t bytes) {
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes > ctx->pin_io_bytes) {
        JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pin_io) JEA_HIP(ctx, hipHostFree(ctx->pin_io));
        ctx->pin_io = nullptr;
        ctx->pin_io_bytes = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        JEA_HIP(ctx, hipHostMalloc(&ctx->pin_io, want, hipHostMallocDefault));
        ctx->pin_io_bytes = want;
    }
    for (hipEvent_t& e : ctx->piece_done)
        if (!e) JEA_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return JEA_OK;
}

// memcpy by a few threads: one thread moves some 10 GB/s, a PCIe link five times that
// (four threads reach 75 GB/s between pageable and pinned memory on the hosts measured, eight little more, sixteen less:
// tools/host_copy_rates.hip)
void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    constexpr size_t kThreads = 4, kMinPerThread = 1u << 20;
    if (bytes < 2 * kMinPerThread) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t piece = ((bytes + kThreads - 1) / kThreads + 63) & ~(size_t)63;
    std::thread workers[kThreads - 1];
    for (size_t t = 1; t < kThreads; ++t) {
        const size_t at = t * piece, len = at < bytes ? (bytes - at < piece ? bytes - at : piece) : 0;
        workers[t - 1] = std::thread([=] { if (len) std::memcpy((char*)dst + at, (const char*)src + at, len); });
    }
    std::memcpy(dst, src, piece < bytes ? piece : bytes);
    for (std::thread& w : workers) w.join();
}

// Host to device through the pinned buffer in pieces: while one piece travels, the next is copied in.
int upload_pipelined(jea_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    int rc;
    if ((rc = ensure_pin_io(ctx, bytes))) return rc;
    constexpr size_t kPieces = 4;
    const size_t piece = ((bytes + kPieces - 1) / kPieces + 4095) & ~(size_t)4095;
    for (size_t at = 0; at < bytes; at += piece) {
        const size_t len = bytes - at < piece ? bytes - at : piece;
        parallel_memcpy((char*)ctx->pin_io + at, (const char*)h_src + at, len);
        JEA_HIP(ctx, hipMemcpyAsync((char*)d_dst + at, (char*)ctx->pin_io + at, len, hipMemcpyHostToDevice, ctx->stream));
    }
    return JEA_OK;
}

// Device to host likewise; returns when h_dst is complete.
int download_pipelined(jea_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    int rc;
    if ((rc = ensure_pin_io(ctx, bytes))) return rc;
    constexpr size_t kPieces = 2;   // measured: two pieces beat four on the way back
    const size_t piece = ((bytes + kPieces - 1) / kPieces + 4095) & ~(size_t)4095;
    size_t n = 0;
    for (size_t at = 0; at < bytes; at += piece, ++n) {
        const size_t len = bytes - at < piece ? bytes - at : piece;
        JEA_HIP(ctx, hipMemcpyAsync((char*)ctx->pin_io + at, (const char*)d_src + at, len, hipMemcpyDeviceToHost, ctx->stream));
        JEA_HIP(ctx, hipEventRecord(ctx->piece_done[n], ctx->stream));
    }
    n = 0;
    for (size_t at = 0; at < bytes; at += piece, ++n) {
        const size_t len = bytes - at < piece ? bytes - at : piece;
        JEA_HIP(ctx, hipEventSynchronize(ctx->piece_done[n]));
        parallel_memcpy((char*)h_dst + at, (const char*)ctx->pin_io + at, len);
    }
    return JEA_OK;
}

void free_buffer(DeviceBuffer& b) {
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

// Canonical code assignment (T.81 Annex C) from the DHT-form tables.
void build_codes(const uint8_t bits[16], const uint8_t* vals, uint32_t* out /* indexed by symbol */, int n_out) {
    for (int i = 0; i < n_out; ++i) out[i] = 0;
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) out[vals[k]] = ((uint32_t)l << 16) | code++;
        code <<= 1;
    }
}

// Returns false when the codes longer than 9 bits span more 16-bit prefixes than the slow table holds.
bool build_decode_table(const uint8_t bits[16], const uint8_t* vals, DecodeTable& t) {
    std::memset(&t, 0, sizeof(t));
    t.first_long = 65536;
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) {
            const uint16_t entry = (uint16_t)((l << 8) | vals[k]);
            if (l <= 9) {
                const uint32_t first = code << (9 - l);
                for (uint32_t j = 0; j < (1u << (9 - l)); ++j) t.fast[first + j] = entry;
            } else {
                const uint32_t first = code << (16 - l);
                if (t.first_long == 65536) t.first_long = first;
                if (first + (1u << (16 - l)) - t.first_long > (uint32_t)kDecodeSlowEntries) return false;
                for (uint32_t j = 0; j < (1u << (16 - l)); ++j) t.slow[first + j - t.first_long] = entry;
            }
            ++code;
        }
        code <<= 1;
    }
    return true;
}

// Canonical code of `sym` in a table given as Annex C counts and values; false when the table lacks it.
bool find_code(const uint8_t bits[16], const uint8_t* vals, uint32_t sym, uint32_t& code_out, uint32_t& len_out) {
    uint32_t code = 0;
    int k = 0;
    for (uint32_t l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) {
            if (vals[k] == sym) {
                code_out = code;
                len_out = l;
                return true;
            }
            ++code;
        }
        code <<= 1;
    }
    return false;
}

// The parallel decoder's view of the same code (kernels.h, StepTableT).  eob_code / eob_len: the end-of-block
// code of the AC table of the same class, for the combined entries.
template <class Table>
bool build_step_table(const uint8_t bits[16], const uint8_t* vals, bool is_dc, uint32_t eob_code, uint32_t eob_len,
                      Table& t) {
    constexpr uint32_t F = Table::kFastBits, kSub = 1u << (16 - F);   // sub-table: one entry per 16-bit prefix
    const uint32_t bad = kStepBad | 1u;   // skip one bit, keep going
    for (uint32_t& e : t.fast) e = bad;
    for (uint32_t& e : t.slow) e = bad;
    uint32_t first_bucket = 1u << F;      // first fast index that is resolved through slow[]
    uint32_t code = 0;
    int k = 0;
    for (uint32_t l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) {
            const uint32_t sym = vals[k];
            const uint32_t ss = is_dc ? sym : (sym & 15u), run = is_dc ? 0u : (sym >> 4);
            // entropy_encoder.cpp:786-806, 830-864: sizes above 11 (DC) / 10 (AC) and AC symbols
            // without amplitude other than EOB and ZRL do not exist
            const bool valid = is_dc ? sym <= 11u : (ss <= 10u && (ss != 0u || sym == 0x00u || sym == 0xF0u));
            uint32_t entry = bad;
            if (valid) {
                entry = (l + ss) | (run << 8) | (l << 16) | (((32u - ss) & 31u) << 24);
                if (!is_dc && sym == 0x00u) entry |= kStepEndsBlock;
                if (ss) entry |= kStepHasAmp | (is_dc ? kStepDcAdd : 0u);
            }
            if (l <= F) {
                const uint32_t first = code << (F - l);
                for (uint32_t j = 0; j < (1u << (F - l)); ++j) t.fast[first + j] = entry;
            } else {
                const uint32_t first = code << (16 - l);   // as a 16-bit prefix
                if (first_bucket == (1u << F)) first_bucket = first >> (16 - F);
                // canonical codes grow with their length: everything from here up goes through slow[]
                const uint32_t at = first - (first_bucket << (16 - F));
                if (at + (1u << (16 - l)) > (uint32_t)Table::kSlowEntries) return false;
                for (uint32_t j = 0; j < (1u << (16 - l)); ++j) t.slow[at + j] = entry;
            }
            ++code;
        }
        code <<= 1;
    }
    if (((1u << F) - first_bucket) * kSub > (uint32_t)Table::kSlowEntries) return false;
    // combined entries: the symbol, its amplitude bits and the end-of-block code all inside the index bits
    if (eob_len >= 1 && eob_len <= 7) {
        for (uint32_t j = 0; j < first_bucket; ++j) {
            const uint32_t e = t.fast[j], used = e & 0xffu;
            if ((e & (kStepBad | kStepEndsBlock)) || !(is_dc || (e & kStepHasAmp)) || used + eob_len > F) continue;
            if (((j >> (F - used - eob_len)) & ((1u << eob_len) - 1u)) != eob_code) continue;
            t.fast[j] = (e + eob_len) | kStepEndsBlock | (eob_len << kStepEobLenShift);
        }
    }
    for (uint32_t j = first_bucket; j < (1u << F); ++j) {
        const uint32_t offset = (uint32_t)offsetof(Table, slow) + (j - first_bucket) * kSub * 4u;
        t.fast[j] = ((16u + F) << 8) | (offset << 16);   // sub-tables of 2^(16 - F) entries: the index is the next 16 - F bits
    }
    return true;
}

bool valid_dims(uint16_t w, uint16_t h) { return w >= 1 && h >= 1; }

size_t mcus_of(uint16_t w, uint16_t h) { return (size_t)((w + 7) / 8) * (size_t)((h + 7) / 8); }

QuantTables to_float_tables(const uint16_t lum_q[64], const uint16_t chr_q[64]) {
    QuantTables q;
    for (int i = 0; i < 64; ++i) {
        q.lum[i] = (float)lum_q[i];
        q.chr[i] = (float)chr_q[i];
        const int t = 8 * (i & 7) + (i >> 3);
        q.lum_t[t] = q.lum[i];
        q.chr_t[t] = q.chr[i];
        q.rlum_t[t] = 1.0f / q.lum[i];
        q.rchr_t[t] = 1.0f / q.chr[i];
    }
    return q;
}

bool tables_ok(const uint16_t lum_q[64], const uint16_t chr_q[64]) {
    if (!lum_q || !chr_q) return false;
    for (int i = 0; i < 64; ++i)
        if (lum_q[i] == 0 || chr_q[i] == 0) return false;
    return true;
}

// Scratch of one entropy-encode pass.
// `for_stripe`: stripes pack into an unstuffed copy first (their bit offsets come from other contexts).
int prepare_entropy_scratch(jea_ctx* ctx, size_t n_images, size_t mcus, size_t out_capacity, EntropyScratch& sc,
                            bool for_stripe = false) {
    const size_t nbc = mcus * 3, n_tiles = tiles_per_image(mcus);
    int rc;
    if ((rc = ensure(ctx, ctx->len16, n_images * nbc * sizeof(uint16_t)))) return rc;
    if ((rc = ensure(ctx, ctx->slots, n_images * n_tiles * (size_t)kSlotWords * kSlotStride * sizeof(uint32_t))))
        return rc;
    if ((rc = ensure(ctx, ctx->tile_bits, n_images * n_tiles * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->tile_base, n_images * n_tiles * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->tile_ffbase, n_images * n_tiles * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->img_words, 3 * n_images * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->tile_ff8, n_images * n_tiles * 4 * sizeof(uint32_t)))) return rc;
    size_t ucap = 0;
    if (for_stripe) {
        // The packed stream of an image is never longer than its stuffed one.
        const size_t worst = n_images * ((nbc * 1660 + 7) / 8 + 16);
        ucap = (out_capacity + 16 * n_images < worst) ? out_capacity + 16 * n_images : worst;
        if ((rc = ensure(ctx, ctx->unstuffed, ucap))) return rc;
        if ((rc = ensure(ctx, ctx->tile_ff, n_images * n_tiles * sizeof(uint32_t)))) return rc;
    }
    sc.len16 = (uint16_t*)ctx->len16.ptr;
    sc.chunk_masks = nullptr;  // set by callers whose coefficients come with masks from the forward kernel
    sc.planar = false;
    sc.mode = ctx->mode;
    sc.slots = (uint32_t*)ctx->slots.ptr;
    sc.tile_bits = (uint32_t*)ctx->tile_bits.ptr;
    sc.tile_base = (uint64_t*)ctx->tile_base.ptr;
    sc.tile_ff = (uint32_t*)ctx->tile_ff.ptr;
    sc.tile_ffbase = (uint64_t*)ctx->tile_ffbase.ptr;
    sc.img_bits = (uint64_t*)ctx->img_words.ptr;
    sc.img_ubase = sc.img_bits + n_images;
    sc.img_ff = sc.img_ubase + n_images;
    sc.unstuffed = (uint8_t*)ctx->unstuffed.ptr;
    sc.unstuffed_capacity = ucap;
    sc.tile_ff8 = (uint32_t*)ctx->tile_ff8.ptr;
    return JEA_OK;
}

int report_status(jea_ctx* ctx, int32_t st, const char* what) {
    switch (st) {
        case JEA_OK: return JEA_OK;
        case JEA_ERR_CAPACITY: return fail(ctx, st, std::string(what) + ": output capacity too small");
        case JEA_ERR_RANGE: return fail(ctx, st, "Invalid runtime encoding encountered.");  // entropy_encoder.cpp:748
        case JEA_ERR_BAD_STREAM: return fail(ctx, st, "Invalid Huffman code encountered in input JPEG data.");
        case JEA_ERR_NO_EOI: return fail(ctx, st, "Failed to find EOI marker");
        default: return fail(ctx, st, std::string(what) + ": device reported status " + std::to_string(st));
    }
}

int finish_status(jea_ctx* ctx, const int32_t* d_status, const char* what) {
    int32_t st = 0;
    JEA_HIP(ctx, hipMemcpyAsync(&st, d_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return report_status(ctx, st, what);
}

struct HeaderCheck {
    const uint8_t* d;
    size_t p = 0;
    unsigned u8() { return d[p++]; }
    unsigned u16() { p += 2; return (unsigned)((d[p - 2] << 8) | d[p - 1]); }
};

}  // namespace

namespace {

// host -> device staging, kernel, device -> host; `in_bytes` / `out_bytes` per call
template <class Launch>
int staged_call(jea_ctx* ctx, const void* h_in, size_t in_bytes, void* h_out, size_t out_bytes, Launch&& launch) {
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stage_in, in_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, out_bytes))) return rc;
    JEA_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    launch(ctx->stage_in.ptr, ctx->stage_out.ptr);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipMemcpyAsync(h_out, ctx->stage_out.ptr, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

bool table_ok(const uint16_t t[64]) {
    if (!t) return false;
    for (int i = 0; i < 64; ++i)
        if (t[i] == 0) return false;
    return true;
}

}  // namespace

extern "C" {

int jea_create(int device_id, jea_ctx** out_ctx) {
    if (!out_ctx) return fail(nullptr, JEA_ERR_INVALID_ARG, "jea_create: out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, JEA_ERR_NO_DEVICE, "jea_create: no HIP device available");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, JEA_ERR_INVALID_ARG, "jea_create: device_id out of range");
    jea_ctx* ctx = new (std::nothrow) jea_ctx();
    if (!ctx) return fail(nullptr, JEA_ERR_HIP, "jea_create: out of host memory");
    ctx->device = device_id;
    auto bail = [&](const char* what, hipError_t e) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        jea_destroy(ctx);
        return fail(nullptr, JEA_ERR_HIP, msg);
    };
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    ctx->stream = ctx->own_stream;
    if ((e = static_scratch_of(device_id, &ctx->static_scratch)) != hipSuccess) return bail("hipGetSymbolAddress", e);

    HuffTables ht;
    build_codes(kDcLumBits, kDcVals, ht.dc[0], 12);
    build_codes(kDcChrBits, kDcVals, ht.dc[1], 12);
    build_codes(kAcLumBits, kAcLumVals, ht.ac[0], 256);
    build_codes(kAcChrBits, kAcChrVals, ht.ac[1], 256);
    DecodeTables dt;
    if (!build_decode_table(kDcLumBits, kDcVals, dt.tab[0]) || !build_decode_table(kAcLumBits, kAcLumVals, dt.tab[1]) ||
        !build_decode_table(kDcChrBits, kDcVals, dt.tab[2]) || !build_decode_table(kAcChrBits, kAcChrVals, dt.tab[3])) {
        jea_destroy(ctx);
        return fail(nullptr, JEA_ERR_INVALID_ARG, "jea_create: Huffman tables do not fit the decode tables");
    }
    StepTables st;
    uint32_t eob_code[2] = {0, 0}, eob_len[2] = {0, 0};
    if (!find_code(kAcLumBits, kAcLumVals, 0x00u, eob_code[0], eob_len[0]) || !find_code(kAcChrBits, kAcChrVals, 0x00u, eob_code[1], eob_len[1]) ||
        !build_step_table(kDcLumBits, kDcVals, true, eob_code[0], eob_len[0], st.cls[0].dc) ||
        !build_step_table(kAcLumBits, kAcLumVals, false, eob_code[0], eob_len[0], st.cls[0].ac) ||
        !build_step_table(kDcChrBits, kDcVals, true, eob_code[1], eob_len[1], st.cls[1].dc) ||
        !build_step_table(kAcChrBits, kAcChrVals, false, eob_code[1], eob_len[1], st.cls[1].ac)) {
        jea_destroy(ctx);
        return fail(nullptr, JEA_ERR_INVALID_ARG, "jea_create: Huffman tables do not fit the decode tables");
    }
    if ((e = hipMalloc((void**)&ctx->d_steps, sizeof(StepTables))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemcpy(ctx->d_steps, &st, sizeof(st), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    if ((e = hipMalloc((void**)&ctx->d_huff, sizeof(HuffTables))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&ctx->d_decode, sizeof(DecodeTables))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&ctx->d_status, 64)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemcpy(ctx->d_huff, &ht, sizeof(ht), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    if ((e = hipMemcpy(ctx->d_decode, &dt, sizeof(dt), hipMemcpyHostToDevice)) != hipSuccess)
        return bail("hipMemcpy", e);
    *out_ctx = ctx;
    return JEA_OK;
}

void jea_destroy(jea_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->lane_stream) (void)hipStreamSynchronize(ctx->lane_stream);   // it works on the scratch freed below
    for (DeviceBuffer* b : {&ctx->coefs, &ctx->masks, &ctx->len16, &ctx->slots, &ctx->tile_bits, &ctx->tile_base, &ctx->tile_ff,
                            &ctx->tile_ffbase, &ctx->img_words, &ctx->unstuffed, &ctx->stage_in, &ctx->stage_out,
                            &ctx->seg_offsets, &ctx->out_offsets, &ctx->out_sizes, &ctx->stream_copy, &ctx->small,
                            &ctx->ustream, &ctx->chunk_drops, &ctx->chunk_base, &ctx->img_drops,
                            &ctx->sub_end, &ctx->sub_used, &ctx->sub_info, &ctx->grp_sum, &ctx->grp_base,
                            &ctx->round_flags, &ctx->grp_pending, &ctx->grp_end, &ctx->sub_guess, &ctx->first_bad, &ctx->ctx_keys,
                            &ctx->huff_hist, &ctx->huff_tables, &ctx->huff_spec, &ctx->step_tables,
                            &ctx->ctx_vals, &ctx->tile_ff8, &ctx->group_flags})
        free_buffer(*b);
    for (hipStream_t st : {ctx->copy_in, ctx->copy_out})
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (PipeSlot& sl : ctx->pipe) {
        free_buffer(sl.d_in);
        free_buffer(sl.d_out);
        free_buffer(sl.d_meta);
        for (void* h : {sl.pin_in, sl.pin_out, sl.pin_meta})
            if (h) (void)hipHostFree(h);
        for (hipEvent_t e : {sl.h2d, sl.work, sl.d2h})
            if (e) (void)hipEventDestroy(e);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pin_io) (void)hipHostFree(ctx->pin_io);
    for (hipEvent_t e : ctx->piece_done)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->marks)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ctx->lane_fork, ctx->lane_join, ctx->lane_offsets[0], ctx->lane_offsets[1]})
        if (e) (void)hipEventDestroy(e);
    if (ctx->lane_stream) (void)hipStreamDestroy(ctx->lane_stream);
    if (ctx->d_huff) (void)hipFree(ctx->d_huff);
    if (ctx->d_decode) (void)hipFree(ctx->d_decode);
    if (ctx->d_steps) (void)hipFree(ctx->d_steps);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* jea_last_error(const jea_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int jea_set_stream(jea_ctx* ctx, void* hip_stream) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return JEA_OK;
}

void* jea_get_stream(jea_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int jea_set_colour_mapper(jea_ctx* ctx, int mapper) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (mapper == JEA_MAPPER_YCBCR) ctx->mode = ComponentMode{1u, 0u};
    else if (mapper == JEA_MAPPER_RGB) ctx->mode = ComponentMode{7u, 1u};
    else return fail(ctx, JEA_ERR_INVALID_ARG, "jea_set_colour_mapper: unknown mapper");
    return JEA_OK;
}

int jea_set_huffman_optimised(jea_ctx* ctx, int on) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    ctx->optimal_huffman = on != 0;
    return JEA_OK;
}

int jea_set_profiling(jea_ctx* ctx, int on) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    if (on)
        for (hipEvent_t& e : ctx->marks)
            if (!e) JEA_HIP(ctx, hipEventCreate(&e));
    ctx->profiling = on != 0;
    ctx->n_marks = 0;
    return JEA_OK;
}

int jea_last_phase_ms(jea_ctx* ctx, float* ms, size_t capacity, size_t* n_phases) {
    if (!ctx || !ms || !n_phases) return JEA_ERR_INVALID_ARG;
    *n_phases = 0;
    if (ctx->n_marks < 2) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_last_phase_ms: no profiled encode or decode call yet");
    if (capacity < (size_t)ctx->n_marks - 1) return fail(ctx, JEA_ERR_CAPACITY, "jea_last_phase_ms: array too short");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipEventSynchronize(ctx->marks[ctx->n_marks - 1]));
    for (int i = 0; i + 1 < ctx->n_marks; ++i) JEA_HIP(ctx, hipEventElapsedTime(&ms[i], ctx->marks[i], ctx->marks[i + 1]));
    *n_phases = (size_t)ctx->n_marks - 1;
    return JEA_OK;
}

int jea_set_scratch_budget(jea_ctx* ctx, size_t bytes) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    ctx->scratch_budget = bytes;
    return JEA_OK;
}

int jea_synchronize(jea_ctx* ctx) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

// ---- tables ---------------------------------------------------------------

void jea_quant_tables(int quality, uint16_t lum_q[64], uint16_t chr_q[64]) {
    if (quality < 1) quality = 1;  // quantiser.cpp:6-11
    else if (quality > 100) quality = 100;
    // :25 -- the scale is formed in integer arithmetic, then held in a float
    const float S = (float)((quality < 50) ? 5000 / quality : 200 - 2 * quality);
    for (int i = 0; i < 64; ++i) {
        // :27-29 / :48-50 -- float arithmetic, floor, narrowing to uint16; :30-32 zero becomes one
        lum_q[i] = (uint16_t)std::floor((S * (float)kBaseLum[i] + 50.0f) / 100.0f);
        chr_q[i] = (uint16_t)std::floor((S * (float)kBaseChr[i] + 50.0f) / 100.0f);
        if (lum_q[i] == 0) lum_q[i] = 1;
        if (chr_q[i] == 0) chr_q[i] = 1;
    }
}

void jea_write_header(uint8_t out[JEA_HEADER_BYTES], uint16_t width, uint16_t height, const uint16_t lum_q[64],
                      const uint16_t chr_q[64]) {
    size_t p = 0;
    auto b = [&](unsigned v) { out[p++] = (uint8_t)v; };
    auto w = [&](unsigned v) { out[p++] = (uint8_t)(v >> 8); out[p++] = (uint8_t)v; };
    w(0xFFD8);                                                  // SOI, encoder.cpp:76
    w(0xFFE0); w(16);                                           // APP0 :79-90
    b('J'); b('F'); b('I'); b('F'); b(0);
    w(0x0102); b(0); w(0x0010); w(0x0010); w(0x0000);
    w(0xFFDB); w(2 + 2 * 65);                                   // DQT, quantiser.cpp:87-102
    b(0x00);
    for (int k = 0; k < 64; ++k) b(lum_q[kZigzag[k]]);          // pushByte keeps the low 8 bits
    b(0x01);
    for (int k = 0; k < 64; ++k) b(chr_q[kZigzag[k]]);
    w(0xFFC0); w(17); b(8); w(height); w(width); b(3);          // SOF0, encoder.cpp:99-116
    b(1); b(0x11); b(0);
    b(2); b(0x11); b(1);
    b(3); b(0x11); b(1);
    w(0xFFC4); w(2 + 4 * 17 + 12 + 12 + 162 + 162);             // DHT, entropy_encoder.cpp:672-693
    b(0x00); for (int i = 0; i < 16; ++i) b(kDcLumBits[i]); for (int i = 0; i < 12; ++i) b(kDcVals[i]);
    b(0x10); for (int i = 0; i < 16; ++i) b(kAcLumBits[i]); for (int i = 0; i < 162; ++i) b(kAcLumVals[i]);
    b(0x01); for (int i = 0; i < 16; ++i) b(kDcChrBits[i]); for (int i = 0; i < 12; ++i) b(kDcVals[i]);
    b(0x11); for (int i = 0; i < 16; ++i) b(kAcChrBits[i]); for (int i = 0; i < 162; ++i) b(kAcChrVals[i]);
    w(0xFFDA); w(12); b(3);                                     // SOS, encoder.cpp:122-137
    b(1); b(0x00); b(2); b(0x11); b(3); b(0x11);
    w(0x003F); b(0x00);
}

size_t jea_max_jpeg_bytes(uint16_t width, uint16_t height) {
    // 1660 bits per block-channel at most (22-bit DC + 63 x 26-bit AC); stuffing can double it
    return JEA_HEADER_BYTES + 2 * ((mcus_of(width, height) * 3 * 1660 + 7) / 8) + 3;
}

int jea_read_header_ex(const uint8_t* h_jpeg, size_t len, uint16_t* width, uint16_t* height, uint16_t lum_q[64],
                       uint16_t chr_q[64], int* own_huffman_tables) {
    if (!width || !height || !lum_q || !chr_q || !own_huffman_tables) return JEA_ERR_INVALID_ARG;
    const int rc = jea_check_header(h_jpeg, len, width, height, nullptr);
    if (rc) return rc;
    // DQT segment at byte 20: marker, length, then {table id, 64 entries in zig-zag order} twice
    const uint8_t* dqt = h_jpeg + 20;
    if (dqt[0] != 0xFF || dqt[1] != 0xDB || dqt[4] != 0x00 || dqt[4 + 65] != 0x01) return JEA_ERR_BAD_HEADER;
    for (int k = 0; k < 64; ++k) {
        lum_q[kZigzag[k]] = dqt[5 + k];
        chr_q[kZigzag[k]] = dqt[5 + 65 + k];
        if (dqt[5 + k] == 0 || dqt[5 + 65 + k] == 0) return JEA_ERR_BAD_HEADER;
    }
    // DHT segment at byte 173 (420 bytes: marker, length, four tables).  Either the tables the reference writes
    // (entropy_encoder.cpp:614-694), or per-image ones in the same layout: the same four ids in the same order,
    // 12 and 162 values, DC codes of at most 11 bits.
    uint8_t own[JEA_HEADER_BYTES];
    jea_write_header(own, *width, *height, lum_q, chr_q);
    if (memcmp(own + 173, h_jpeg + 173, 4) != 0) return JEA_ERR_BAD_HEADER;
    *own_huffman_tables = memcmp(own + kHuffSpecOffset, h_jpeg + kHuffSpecOffset, kHuffSpecBytes) != 0 ? 1 : 0;
    if (*own_huffman_tables) {
        const size_t at[4] = {0, 29, 208, 237};
        for (int t = 0; t < 4; ++t) {
            const uint8_t* tab = h_jpeg + kHuffSpecOffset + at[t];
            if (tab[0] != own[kHuffSpecOffset + at[t]]) return JEA_ERR_BAD_HEADER;
            unsigned n = 0, code = 0;
            for (int l = 1; l <= 16; ++l) {
                n += tab[l];
                code = (code + tab[l]) << 1;
                if (code > (2u << l)) return JEA_ERR_BAD_HEADER;          // more codes than the length has
                if (!(t & 1) && l > 11 && tab[l]) return JEA_ERR_BAD_HEADER;
            }
            if (n != ((t & 1) ? 162u : 12u)) return JEA_ERR_BAD_HEADER;
        }
    }
    return JEA_OK;
}

int jea_read_header(const uint8_t* h_jpeg, size_t len, uint16_t* width, uint16_t* height, uint16_t lum_q[64],
                    uint16_t chr_q[64]) {
    // The decoder's built-in tables are the ones the reference writes; a file that declares others cannot be
    // decoded by them, and the reference, which skips the segment (encoder.cpp:240), would decode it wrongly.
    int own_tables = 0;
    const int rc = jea_read_header_ex(h_jpeg, len, width, height, lum_q, chr_q, &own_tables);
    if (rc) return rc;
    return own_tables ? JEA_ERR_BAD_HEADER : JEA_OK;
}

int jea_check_header(const uint8_t* h_jpeg, size_t len, uint16_t* hdr_width, uint16_t* hdr_height,
                     const char** msg) {
    const char* m = nullptr;
    uint16_t hw = 0, hh = 0;
    // Field order and messages: encoder.cpp:141-281.
    if (!h_jpeg || len < JEA_HEADER_BYTES) {
        m = "Failed to find SOI marker";
    } else {
        HeaderCheck r{h_jpeg};
#define NEED(cond, text) if (!m && !(cond)) m = text
        NEED(r.u16() == 0xFFD8, "Failed to find SOI marker");
        NEED(r.u16() == 0xFFE0, "Failed to find APP-0 marker");
        {
            const size_t start = r.p;
            const unsigned l = r.u16();
            NEED(r.u8() == 'J', "Failed to find 'J'(FIF) in APP-0 payload");
            NEED(r.u8() == 'F', "Failed to find (J)'F'(IF) in APP-0 payload");
            NEED(r.u8() == 'I', "Failed to find (JF)'I'(F) in APP-0 payload");
            NEED(r.u8() == 'F', "Failed to find (JFI)'F' in APP-0 payload");
            NEED(r.u8() == 0, "Failed to find 'JFIF' null-terminator in APP-0 payload");
            NEED(r.u16() == 0x0102, "JFIF version indicated by APP-0 payload is not 1.02");
            NEED(r.u8() == 0, "Failed to find density units in APP-0 payload");
            NEED(r.u16() == 0x0010, "Failed to find Xdensity in APP-0 payload");
            NEED(r.u16() == 0x0010, "Failed to find Ydensity in APP-0 payload");
            NEED(r.u16() == 0x0000, "Failed to find thumbnail size in APP-0 payload");
            NEED(l == r.p - start, "APP-0 length parameter does not correspond to payload size");
        }
        r.p += 2 + 2 + 2 * 65;  // DQT skipped, :188
        NEED(r.u16() == 0xFFC0, "Failed to find SOF0 marker");
        {
            const size_t start = r.p;
            const unsigned l = r.u16();
            NEED(r.u8() == 8, "Failed to find precision in SOF0 payload");
            hh = (uint16_t)r.u16();
            hw = (uint16_t)r.u16();
            NEED(r.u8() == 3, "Failed to find number of components in SOF0 payload");
            NEED(r.u8() == 1, "Failed to find first component ID in SOF0 payload");
            NEED(r.u8() == 0x11, "Failed to find first component sampling factors in SOF0 payload");
            NEED(r.u8() == 0, "Failed to find first component quantisation table ID in SOF0 payload");
            NEED(r.u8() == 2, "Failed to find second component ID in SOF0 payload");
            NEED(r.u8() == 0x11, "Failed to find second component sampling factors in SOF0 payload");
            NEED(r.u8() == 1, "Failed to find second component quantisation table ID in SOF0 payload");
            NEED(r.u8() == 3, "Failed to find third component ID in SOF0 payload");
            NEED(r.u8() == 0x11, "Failed to find third component sampling factors in SOF0 payload");
            NEED(r.u8() == 1, "Failed to find third component quantisation table ID in SOF0 payload");
            NEED(l == r.p - start, "SOF0 length parameter does not correspond to payload size");
        }
        r.p += 2 + 2 + 4 + 412;  // DHT skipped, :240
        NEED(r.u16() == 0xFFDA, "Failed to find SOS marker");
        {
            const size_t start = r.p;
            const unsigned l = r.u16();
            NEED(r.u8() == 3, "Failed to find number of components in SOS payload");
            NEED(r.u8() == 1, "Failed to find first component ID in SOS payload");
            NEED(r.u8() == 0x00, "Failed to find first component Huffman table ID in SOS payload");
            NEED(r.u8() == 2, "Failed to find second component ID in SOS payload");
            NEED(r.u8() == 0x11, "Failed to find second component Huffman table ID in SOS payload");
            NEED(r.u8() == 3, "Failed to find third component ID in SOS payload");
            NEED(r.u8() == 0x11, "Failed to find third component Huffman table ID in SOS payload");
            NEED(r.u16() == 0x003F, "Failed to find spectral alignment in SOS payload");
            NEED(r.u8() == 0, "Failed to find skip byte in SOS payload");
            NEED(l == r.p - start, "SOS length parameter does not correspond to payload size");
        }
#undef NEED
    }
    if (hdr_width) *hdr_width = hw;
    if (hdr_height) *hdr_height = hh;
    if (msg) *msg = m;
    return m ? JEA_ERR_BAD_HEADER : JEA_OK;
}

// ---- stage entry points -----------------------------------------------------

int jea_fdct_quant_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                         const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_coefs) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb || !d_coefs || n_images == 0 || n_images > 65535 || !valid_dims(width, height) ||
        !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_fdct_quant: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    launch_fdct_quant(ctx->stream, d_rgb, n_images, width, height, to_float_tables(lum_q, chr_q), d_coefs, nullptr, nullptr,
                      false, ctx->mode);
    JEA_HIP(ctx, hipGetLastError());
    return JEA_OK;
}

int jea_fdct_quant_planar_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width,
                                uint16_t height, const uint16_t lum_q[64], const uint16_t chr_q[64],
                                int16_t* d_coefs_planar, uint8_t* d_chunk_masks) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb || !d_coefs_planar || !d_chunk_masks || n_images == 0 || n_images > 65535 ||
        !valid_dims(width, height) || !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_fdct_quant_planar_async: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    launch_fdct_quant(ctx->stream, d_rgb, n_images, width, height, to_float_tables(lum_q, chr_q), d_coefs_planar,
                      nullptr, d_chunk_masks, true, ctx->mode);
    JEA_HIP(ctx, hipGetLastError());
    return JEA_OK;
}

int jea_fdct_quant(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                   const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_coefs) {
    const int rc = jea_fdct_quant_async(ctx, d_rgb, n_images, width, height, lum_q, chr_q, d_coefs);
    if (rc) return rc;
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

int jea_idct_dequant(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, uint16_t width, uint16_t height,
                     const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_rgb) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb || !d_coefs || n_images == 0 || n_images > 65535 || !valid_dims(width, height) ||
        !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_idct_dequant: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    launch_idct_dequant(ctx->stream, d_coefs, n_images, width, height, to_float_tables(lum_q, chr_q), d_rgb, nullptr,
                        ctx->mode);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

int jea_rgb_to_ycbcr(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_pixels, uint8_t* d_ycc) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb || !d_ycc || n_pixels == 0) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_rgb_to_ycbcr: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    launch_rgb_to_ycbcr(ctx->stream, d_rgb, n_pixels, d_ycc);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

int jea_ycbcr_to_rgb(jea_ctx* ctx, const uint8_t* d_ycc, size_t n_pixels, uint8_t* d_rgb) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb || !d_ycc || n_pixels == 0) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_ycbcr_to_rgb: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    launch_ycbcr_to_rgb(ctx->stream, d_ycc, n_pixels, d_rgb);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

int jea_block_bit_lengths(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image,
                          uint16_t* d_lengths) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_coefs || !d_lengths || n_images == 0 || n_images > 65535 || mcus_per_image == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_block_bit_lengths: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->tile_bits, n_images * tiles_per_image(mcus_per_image) * sizeof(uint32_t)))) return rc;
    JEA_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
    launch_bit_lengths(ctx->stream, d_coefs, n_images, mcus_per_image, ctx->d_huff, d_lengths,
                       (uint32_t*)ctx->tile_bits.ptr, ctx->d_status, ctx->mode);
    JEA_HIP(ctx, hipGetLastError());
    return finish_status(ctx, ctx->d_status, "jea_block_bit_lengths");
}

static int entropy_encode_common(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, size_t mcus,
                                 const HeaderBytes& header, int header_len, bool with_eoi, uint8_t* d_out,
                                 size_t out_capacity, uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status,
                                 bool status_is_clear, const uint8_t* d_chunk_masks = nullptr, bool planar = false) {
    EntropyScratch sc;
    int rc = prepare_entropy_scratch(ctx, n_images, mcus, out_capacity, sc);
    if (rc) return rc;
    sc.chunk_masks = d_chunk_masks;
    sc.planar = planar;
    if (!status_is_clear) JEA_HIP(ctx, hipMemsetAsync(d_status, 0, sizeof(int32_t), ctx->stream));
    launch_entropy_encode(ctx->stream, d_coefs, n_images, mcus, ctx->d_huff, sc, header, header_len, with_eoi, d_out,
                          out_capacity, d_offsets, d_sizes, d_status);
    JEA_HIP(ctx, hipGetLastError());
    return JEA_OK;
}

static int fetch_offsets_sizes(jea_ctx* ctx, size_t n_images, const uint64_t* d_offsets, const uint64_t* d_sizes,
                               uint64_t* h_offsets, uint64_t* h_sizes) {
    JEA_HIP(ctx, hipMemcpyAsync(h_offsets, d_offsets, n_images * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(h_sizes, d_sizes, n_images * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    return JEA_OK;
}

int jea_entropy_encode(jea_ctx* ctx, const int16_t* d_coefs, size_t n_images, size_t mcus_per_image, uint8_t* d_out,
                       size_t out_capacity, uint64_t* h_offsets, uint64_t* h_sizes) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_coefs || !d_out || !h_offsets || !h_sizes || n_images == 0 || n_images > 65535 || mcus_per_image == 0 ||
        out_capacity == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_entropy_encode: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->out_offsets, n_images * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->out_sizes, n_images * sizeof(uint64_t)))) return rc;
    HeaderBytes hb{};
    rc = entropy_encode_common(ctx, d_coefs, n_images, mcus_per_image, hb, 0, false, d_out, out_capacity,
                               (uint64_t*)ctx->out_offsets.ptr, (uint64_t*)ctx->out_sizes.ptr, ctx->d_status, false);
    if (rc) return rc;
    if ((rc = fetch_offsets_sizes(ctx, n_images, (uint64_t*)ctx->out_offsets.ptr, (uint64_t*)ctx->out_sizes.ptr,
                                  h_offsets, h_sizes)))
        return rc;
    return finish_status(ctx, ctx->d_status, "jea_entropy_encode");
}

// ---- whole encode -------------------------------------------------------------

// The launches of a whole encode, given scratch that is large enough: forward kernel into chunk-planar
// coefficients (all-zero chunks not even written), then the entropy kernels.
// `lane`: null = on the context's stream, the pass clearing the status word if it is the call's first; else the
// stream of a pass of a batch that runs on two streams (the caller has cleared the status word).
static int enqueue_encode(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                          const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_coefs, uint8_t* d_masks,
                          EntropyScratch sc, uint8_t* d_out, size_t out_capacity, uint64_t* d_offsets,
                          uint64_t* d_sizes, int32_t* d_status, const uint64_t* d_prev_offset = nullptr,
                          const uint64_t* d_prev_size = nullptr, hipStream_t lane = nullptr) {
    const size_t mcus = mcus_of(width, height);
    HeaderBytes hb{};
    jea_write_header(hb.b, width, height, lum_q, chr_q);
    const hipStream_t stream = lane ? lane : ctx->stream;
    const bool timed = ctx->profiling && !lane;
    ctx->n_marks = 0;
    if (timed) JEA_HIP(ctx, hipEventRecord(ctx->marks[0], stream));
    // the first pass of a call clears the status word; later passes (of a batch encoded in pieces) only raise it
    launch_fdct_quant(stream, d_rgb, n_images, width, height, to_float_tables(lum_q, chr_q), d_coefs,
                      (d_prev_offset == nullptr && !lane) ? d_status : nullptr, d_masks, true, ctx->mode);
    JEA_HIP(ctx, hipGetLastError());
    if (timed) JEA_HIP(ctx, hipEventRecord(ctx->marks[1], stream));
    sc.chunk_masks = d_masks;
    sc.planar = true;
    sc.marks = timed ? ctx->marks + 2 : nullptr;
    launch_entropy_encode(stream, d_coefs, n_images, mcus, ctx->d_huff, sc, hb, JEA_HEADER_BYTES, true, d_out,
                          out_capacity, d_offsets, d_sizes, d_status, d_prev_offset, d_prev_size);
    JEA_HIP(ctx, hipGetLastError());
    if (timed) {
        JEA_HIP(ctx, hipEventRecord(ctx->marks[3], stream));
        ctx->n_marks = 4;   // forward transform | bit strings | scans, pack and stuff
    }
    return JEA_OK;
}

static bool encode_args_ok(const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                           const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_out, size_t out_capacity,
                           uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status) {
    return d_rgb && d_out && d_offsets && d_sizes && d_status && n_images != 0 && n_images <= 65535 &&
           valid_dims(width, height) && tables_ok(lum_q, chr_q) && out_capacity >= JEA_HEADER_BYTES + 3;
}

// Bytes of scratch an encode of n_images pictures of `mcus` MCUs each takes: the same buffers that
// the whole-encode entry points use (per context, or out of `scratch`), each starting at a multiple of 256 bytes.
namespace {
struct StaticLayout {
    size_t coefs, masks, len16, slots, tile_bits, tile_base, tile_ffbase, img_words, tile_ff8, total;
};
StaticLayout static_layout(size_t n_images, size_t mcus) {
    const size_t nbc = mcus * 3, n_tiles = tiles_per_image(mcus);
    StaticLayout l{};
    size_t at = 0;
    auto take = [&at](size_t bytes) {
        const size_t here = at;
        at += (bytes + 255) & ~(size_t)255;
        return here;
    };
    l.coefs = take(n_images * nbc * 64 * sizeof(int16_t));
    l.masks = take(n_images * nbc);
    l.len16 = take(n_images * nbc * sizeof(uint16_t));
    l.slots = take(n_images * n_tiles * (size_t)kSlotWords * kSlotStride * sizeof(uint32_t));
    l.tile_bits = take(n_images * n_tiles * sizeof(uint32_t));
    l.tile_base = take(n_images * n_tiles * sizeof(uint64_t));
    l.tile_ffbase = take(n_images * n_tiles * sizeof(uint64_t));
    l.img_words = take(3 * n_images * sizeof(uint64_t));
    l.tile_ff8 = take(n_images * n_tiles * 4 * sizeof(uint32_t));
    l.total = at;
    return l;
}
}  // namespace

int jea_encode_batch_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                           const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_out, size_t out_capacity,
                           uint64_t* d_offsets, uint64_t* d_sizes, int32_t* d_status) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!encode_args_ok(d_rgb, n_images, width, height, lum_q, chr_q, d_out, out_capacity, d_offsets, d_sizes, d_status))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_encode_batch: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mcus = mcus_of(width, height);
    // Scratch per image: coefficients (6 B/px) and the worst-case bit strings (53 248 B per tile, 9.9 B/px)
    // dominate.  A batch whose scratch would exceed the context's budget goes through in passes of as many
    // images as fit, one after the other on the stream with the same scratch; each pass continues the
    // file offsets where the one before stopped, so the output is the same as that of one pass.
    const size_t per_image = static_layout(1, mcus).total;
    size_t pass = ctx->scratch_budget / per_image;
    if (pass < 1) pass = 1;
    if (pass > n_images) pass = n_images;
    // (a large batch runs as pieces of kLanePiece images on two streams, see below: scratch for two pieces)
    constexpr size_t kLanePiece = 64;
    // (only on the context's own stream: a caller's stream may have another priority, and streams of different
    // priorities that wait for each other's events are what once stopped a GPU in our experiments)
    const bool two_lanes = pass >= 2 * kLanePiece && n_images >= 2 * kLanePiece && !ctx->profiling && !ctx->optimal_huffman &&
                           ctx->stream == ctx->own_stream;
    if (two_lanes) pass = 2 * kLanePiece;
    int rc;
    if ((rc = ensure(ctx, ctx->coefs, pass * mcus * 3 * 64 * sizeof(int16_t)))) return rc;
    if ((rc = ensure(ctx, ctx->masks, pass * mcus * 3))) return rc;
    // scratch first: growing a buffer synchronises the stream, which must not happen between the launches
    EntropyScratch sc;
    if ((rc = prepare_entropy_scratch(ctx, pass, mcus, out_capacity, sc))) return rc;
    if (ctx->optimal_huffman) {
        if ((rc = ensure(ctx, ctx->huff_hist, pass * 1024 * sizeof(uint32_t)))) return rc;
        if ((rc = ensure(ctx, ctx->huff_tables, pass * sizeof(HuffTables)))) return rc;
        if ((rc = ensure(ctx, ctx->huff_spec, pass * kHuffSpecBytes))) return rc;
        sc.optimal.hist = (uint32_t*)ctx->huff_hist.ptr;
        sc.optimal.tables = (HuffTables*)ctx->huff_tables.ptr;
        sc.optimal.spec = (uint8_t*)ctx->huff_spec.ptr;
    }
    const size_t image_bytes = (size_t)3 * width * height;
    // A large batch goes through as pieces of kLanePiece images that alternate over two streams, each with scratch of
    // its own: the forward transform of one piece (bound by vector-instruction issue) runs beside the entropy kernels
    // of the piece before it (bound by latency).  Only the kernels that place a piece's files in the output wait, for
    // the offsets of the piece before (an event recorded once those are final).  1024 1080p frames at q=90: 8.6 ms in one pass, 8.0 ms as pieces of 64 one after
    // the other, 7.6 ms on two streams (three streams, or pieces of 32 or 128: no better).
    if (two_lanes) {
        if (ctx->lane_stream == nullptr) {
            JEA_HIP(ctx, hipStreamCreateWithFlags(&ctx->lane_stream, hipStreamNonBlocking));
            JEA_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_fork, hipEventDisableTiming));
            JEA_HIP(ctx, hipEventCreateWithFlags(&ctx->lane_join, hipEventDisableTiming));
            for (hipEvent_t& e : ctx->lane_offsets) JEA_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const size_t nbc = mcus * 3, n_tiles = tiles_per_image(mcus);
        JEA_HIP(ctx, hipMemsetAsync(d_status, 0, sizeof(int32_t), ctx->stream));
        JEA_HIP(ctx, hipEventRecord(ctx->lane_fork, ctx->stream));
        JEA_HIP(ctx, hipStreamWaitEvent(ctx->lane_stream, ctx->lane_fork, 0));
        size_t piece = 0;
        for (size_t first = 0; first < n_images; first += kLanePiece, ++piece) {
            const size_t n = n_images - first < kLanePiece ? n_images - first : kLanePiece;
            const size_t lane = piece & 1, at = lane * kLanePiece;   // the lane's share of the scratch: images at ..
            EntropyScratch ls = sc;
            ls.len16 = sc.len16 + at * nbc;
            ls.slots = sc.slots + at * n_tiles * (size_t)kSlotWords * kSlotStride;
            ls.tile_bits = sc.tile_bits + at * n_tiles;
            ls.tile_base = sc.tile_base + at * n_tiles;
            ls.tile_ffbase = sc.tile_ffbase + at * n_tiles;
            ls.tile_ff8 = sc.tile_ff8 + at * n_tiles * 4;
            ls.img_bits = sc.img_bits + 3 * at;
            ls.img_ubase = ls.img_bits + kLanePiece;
            ls.img_ff = ls.img_ubase + kLanePiece;
            ls.wait_before_offsets = piece ? ctx->lane_offsets[lane ^ 1] : nullptr;
            ls.record_after_offsets = ctx->lane_offsets[lane];
            const hipStream_t stream = lane ? ctx->lane_stream : ctx->stream;
            rc = enqueue_encode(ctx, d_rgb + first * image_bytes, n, width, height, lum_q, chr_q,
                                (int16_t*)ctx->coefs.ptr + at * nbc * 64, (uint8_t*)ctx->masks.ptr + at * nbc, ls, d_out,
                                out_capacity, d_offsets + first, d_sizes + first, d_status,
                                first ? d_offsets + first - 1 : nullptr, first ? d_sizes + first - 1 : nullptr, stream);
            if (rc) return rc;
        }
        // the context's stream ends the call: it waits for the other stream's last piece
        JEA_HIP(ctx, hipEventRecord(ctx->lane_join, ctx->lane_stream));
        JEA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_join, 0));
        return JEA_OK;
    }
    for (size_t first = 0; first < n_images; first += pass) {
        const size_t n = n_images - first < pass ? n_images - first : pass;
        rc = enqueue_encode(ctx, d_rgb + first * image_bytes, n, width, height, lum_q, chr_q, (int16_t*)ctx->coefs.ptr,
                            (uint8_t*)ctx->masks.ptr, sc, d_out, out_capacity, d_offsets + first, d_sizes + first, d_status,
                            first ? d_offsets + first - 1 : nullptr, first ? d_sizes + first - 1 : nullptr);
        if (rc) return rc;
    }
    return JEA_OK;
}

size_t jea_static_scratch_needed(size_t n_images, uint16_t width, uint16_t height) {
    if (n_images == 0 || !valid_dims(width, height)) return 0;
    return static_layout(n_images, mcus_of(width, height)).total;
}

int jea_encode_batch_static_async(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width,
                                  uint16_t height, const uint16_t lum_q[64], const uint16_t chr_q[64],
                                  uint8_t* d_out, size_t out_capacity, uint64_t* d_offsets, uint64_t* d_sizes,
                                  int32_t* d_status) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!encode_args_ok(d_rgb, n_images, width, height, lum_q, chr_q, d_out, out_capacity, d_offsets, d_sizes, d_status))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_encode_batch_static_async: invalid argument");
    if (ctx->optimal_huffman)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_encode_batch_static_async: per-image Huffman tables are not available here; use jea_encode_batch_async");
    const size_t mcus = mcus_of(width, height);
    const StaticLayout l = static_layout(n_images, mcus);
    if (l.total > kStaticScratchBytes)
        return fail(ctx, JEA_ERR_CAPACITY,
                    "jea_encode_batch_static_async: " + std::to_string(n_images) + " image(s) of " +
                        std::to_string(width) + "x" + std::to_string(height) + " need " + std::to_string(l.total) +
                        " bytes of scratch, the built-in buffer has " + std::to_string(kStaticScratchBytes) +
                        "; use jea_encode_batch_async");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    unsigned char* const base = ctx->static_scratch;   // this GPU's copy, found in jea_create
    EntropyScratch sc{};
    sc.len16 = (uint16_t*)(base + l.len16);
    sc.mode = ctx->mode;
    sc.slots = (uint32_t*)(base + l.slots);
    sc.tile_bits = (uint32_t*)(base + l.tile_bits);
    sc.tile_base = (uint64_t*)(base + l.tile_base);
    sc.tile_ffbase = (uint64_t*)(base + l.tile_ffbase);
    sc.img_bits = (uint64_t*)(base + l.img_words);
    sc.img_ubase = sc.img_bits + n_images;
    sc.img_ff = sc.img_ubase + n_images;
    sc.tile_ff8 = (uint32_t*)(base + l.tile_ff8);
    // tile_ff and unstuffed stay null: only the stripe kernels use them
    // Every context and stream of this GPU shares the buffer: this call's kernels start after those of the call
    // that used it last (an event wait, no allocation, no host wait), and the pair wait / record is taken under a
    // lock so that two host threads cannot both wait for the same predecessor.  A stream under graph capture is
    // left alone: an event recorded into a capture cannot order work outside it, so there the graph's owner
    // keeps its launches apart.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capture) != hipSuccess) {
        (void)hipGetLastError();
        capture = hipStreamCaptureStatusNone;
    }
    if (capture != hipStreamCaptureStatusNone)
        return enqueue_encode(ctx, d_rgb, n_images, width, height, lum_q, chr_q, (int16_t*)(base + l.coefs),
                              base + l.masks, sc, d_out, out_capacity, d_offsets, d_sizes, d_status);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    JEA_HIP(ctx, hipStreamWaitEvent(ctx->stream, g_scratch_free[ctx->device], 0));
    const int rc = enqueue_encode(ctx, d_rgb, n_images, width, height, lum_q, chr_q, (int16_t*)(base + l.coefs),
                                  base + l.masks, sc, d_out, out_capacity, d_offsets, d_sizes, d_status);
    JEA_HIP(ctx, hipEventRecord(g_scratch_free[ctx->device], ctx->stream));
    return rc;
}

size_t jea_static_scratch_bytes(void) { return kStaticScratchBytes; }

int jea_encode_batch(jea_ctx* ctx, const uint8_t* d_rgb, size_t n_images, uint16_t width, uint16_t height,
                     const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* d_out, size_t out_capacity,
                     uint64_t* h_offsets, uint64_t* h_sizes) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_offsets || !h_sizes || n_images == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_encode_batch: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->out_offsets, n_images * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->out_sizes, n_images * sizeof(uint64_t)))) return rc;
    uint64_t* d_offsets = (uint64_t*)ctx->out_offsets.ptr;
    uint64_t* d_sizes = (uint64_t*)ctx->out_sizes.ptr;
    rc = jea_encode_batch_async(ctx, d_rgb, n_images, width, height, lum_q, chr_q, d_out, out_capacity, d_offsets,
                                d_sizes, ctx->d_status);
    if (rc) return rc;
    if ((rc = fetch_offsets_sizes(ctx, n_images, d_offsets, d_sizes, h_offsets, h_sizes))) return rc;
    return finish_status(ctx, ctx->d_status, "jea_encode_batch");
}

int jea_encode_host(jea_ctx* ctx, const uint8_t* h_rgb, uint16_t width, uint16_t height, const uint16_t lum_q[64],
                    const uint16_t chr_q[64], uint8_t* h_out, size_t out_capacity, size_t* out_len) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_rgb || !h_out || !out_len || !valid_dims(width, height) || out_capacity < JEA_HEADER_BYTES + 3)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_encode_host: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)3 * width * height;
    const size_t worst = jea_max_jpeg_bytes(width, height);
    const size_t cap = out_capacity < worst ? out_capacity : worst;
    int rc;
    if ((rc = ensure(ctx, ctx->stage_in, in_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, cap))) return rc;
    if ((rc = upload_pipelined(ctx, ctx->stage_in.ptr, h_rgb, in_bytes))) return rc;
    uint64_t off = 0, size = 0;
    rc = jea_encode_batch(ctx, (const uint8_t*)ctx->stage_in.ptr, 1, width, height, lum_q, chr_q,
                          (uint8_t*)ctx->stage_out.ptr, cap, &off, &size);
    if (rc) return rc;
    if ((rc = download_pipelined(ctx, h_out, (const uint8_t*)ctx->stage_out.ptr + off, size))) return rc;
    *out_len = (size_t)size;
    return JEA_OK;
}

// ---- stripes --------------------------------------------------------------------

int jea_stripe_transform(jea_ctx* ctx, const uint8_t* d_rgb_stripe, uint16_t width, uint16_t stripe_height,
                         const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t last_dc[3]) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb_stripe || !last_dc || !valid_dims(width, stripe_height) || !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_transform: invalid argument");
    if (ctx->optimal_huffman)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_transform: per-image Huffman tables are not available for stripes");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mcus = mcus_of(width, stripe_height);
    int rc;
    if ((rc = ensure(ctx, ctx->coefs, mcus * 3 * 64 * sizeof(int16_t)))) return rc;
    if ((rc = ensure(ctx, ctx->masks, mcus * 3))) return rc;
    // worst-case scratch for the later steps, so that they never reallocate between launches
    EntropyScratch sc;
    const size_t worst = (mcus * 3 * 1660 + 7) / 8 + 64;
    if ((rc = prepare_entropy_scratch(ctx, 1, mcus, worst, sc, true))) return rc;
    int16_t* d_coefs = (int16_t*)ctx->coefs.ptr;
    launch_fdct_quant(ctx->stream, d_rgb_stripe, 1, width, stripe_height, to_float_tables(lum_q, chr_q), d_coefs,
                      ctx->d_status, (uint8_t*)ctx->masks.ptr, false, ctx->mode);
    JEA_HIP(ctx, hipGetLastError());
    for (int ch = 0; ch < 3; ++ch)
        JEA_HIP(ctx, hipMemcpyAsync(&last_dc[ch], d_coefs + ((mcus - 1) * 3 + ch) * 64, sizeof(int16_t),
                                    hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stripe_mcus = mcus;
    ctx->stripe_height = stripe_height;
    ctx->stripe_capacity = worst;
    ctx->stripe_stage = 1;
    return JEA_OK;
}

int jea_stripe_bits(jea_ctx* ctx, const int16_t prev_dc[3], uint64_t* total_bits, uint32_t* head_bits) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!prev_dc || !total_bits || !head_bits) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_bits: invalid argument");
    if (ctx->stripe_stage != 1) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_bits: call jea_stripe_transform first");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    EntropyScratch sc;
    int rc;
    if ((rc = prepare_entropy_scratch(ctx, 1, ctx->stripe_mcus, ctx->stripe_capacity, sc, true))) return rc;
    if ((rc = ensure(ctx, ctx->small, 256))) return rc;
    sc.chunk_masks = (const uint8_t*)ctx->masks.ptr;
    StripeInfo st;
    for (int i = 0; i < 3; ++i) st.initial_dc[i] = prev_dc[i];
    uint32_t* d_head = (uint32_t*)ctx->small.ptr;
    launch_stripe_bits(ctx->stream, (const int16_t*)ctx->coefs.ptr, ctx->stripe_mcus, ctx->d_huff, sc, st, d_head,
                       ctx->d_status);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipMemcpyAsync(total_bits, sc.img_bits, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(head_bits, d_head, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = finish_status(ctx, ctx->d_status, "jea_stripe_bits"))) return rc;
    ctx->stripe_stage = 2;
    return JEA_OK;
}

int jea_stripe_finish(jea_ctx* ctx, uint64_t bits_before, int is_first, int has_next, uint32_t next_head_bits,
                      uint16_t image_width, uint16_t image_height, const uint16_t lum_q[64],
                      const uint16_t chr_q[64], uint8_t* d_segment, size_t capacity, size_t* segment_bytes) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_segment || !segment_bytes || capacity == 0 || !tables_ok(lum_q, chr_q) ||
        !valid_dims(image_width, image_height))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish: invalid argument");
    if (ctx->stripe_stage != 2) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish: call jea_stripe_bits first");
    if (has_next && ctx->stripe_height % 8 != 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish: only the last stripe may have a height that is not a multiple of 8");
    if (is_first && bits_before != 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish: the first stripe starts at bit 0");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    EntropyScratch sc;
    int rc;
    if ((rc = prepare_entropy_scratch(ctx, 1, ctx->stripe_mcus, ctx->stripe_capacity, sc, true))) return rc;
    if ((rc = ensure(ctx, ctx->out_offsets, sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->out_sizes, sizeof(uint64_t)))) return rc;
    StripeInfo st;
    st.bit_base = bits_before;
    st.first_byte = (bits_before + 7) >> 3;
    st.origin_byte = st.first_byte & ~uint64_t(3);
    st.next_head = has_next ? next_head_bits : 0;
    st.has_next = has_next ? 1 : 0;
    st.active = 1;
    HeaderBytes hb{};
    if (is_first) jea_write_header(hb.b, image_width, image_height, lum_q, chr_q);
    launch_stripe_finish(ctx->stream, ctx->stripe_mcus, sc, st, hb, is_first ? JEA_HEADER_BYTES : 0, !has_next, d_segment,
                         capacity, (uint64_t*)ctx->out_offsets.ptr, (uint64_t*)ctx->out_sizes.ptr, ctx->d_status);
    JEA_HIP(ctx, hipGetLastError());
    uint64_t size = 0;
    JEA_HIP(ctx, hipMemcpyAsync(&size, ctx->out_sizes.ptr, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = finish_status(ctx, ctx->d_status, "jea_stripe_finish"))) return rc;
    *segment_bytes = (size_t)size;
    ctx->stripe_stage = 0;
    return JEA_OK;
}

// The same three steps with the facts in device memory: nothing is copied to the host and nothing waits, so a
// rank can enqueue step, exchange, step, exchange, step on one stream.

int jea_stripe_transform_async(jea_ctx* ctx, const uint8_t* d_rgb_stripe, uint16_t width, uint16_t stripe_height,
                               const uint16_t lum_q[64], const uint16_t chr_q[64], int16_t* d_last_dc) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_rgb_stripe || !d_last_dc || !valid_dims(width, stripe_height) || !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_transform_async: invalid argument");
    if (ctx->optimal_huffman)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_transform_async: per-image Huffman tables are not available for stripes");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mcus = mcus_of(width, stripe_height);
    int rc;
    // all scratch of all three steps now (growing a buffer waits for the stream, which the later steps must not)
    if ((rc = ensure(ctx, ctx->coefs, mcus * 3 * 64 * sizeof(int16_t)))) return rc;
    if ((rc = ensure(ctx, ctx->masks, mcus * 3))) return rc;
    EntropyScratch sc;
    const size_t worst = (mcus * 3 * 1660 + 7) / 8 + 64;
    if ((rc = prepare_entropy_scratch(ctx, 1, mcus, worst, sc, true))) return rc;
    if ((rc = ensure(ctx, ctx->small, 512))) return rc;
    if ((rc = ensure(ctx, ctx->out_offsets, sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->out_sizes, sizeof(uint64_t)))) return rc;
    int16_t* d_coefs = (int16_t*)ctx->coefs.ptr;
    launch_fdct_quant(ctx->stream, d_rgb_stripe, 1, width, stripe_height, to_float_tables(lum_q, chr_q), d_coefs,
                      ctx->d_status, (uint8_t*)ctx->masks.ptr, false, ctx->mode);
    launch_stripe_last_dc(ctx->stream, d_coefs, mcus, d_last_dc);
    JEA_HIP(ctx, hipGetLastError());
    ctx->stripe_mcus = mcus;
    ctx->stripe_height = stripe_height;
    ctx->stripe_capacity = worst;
    ctx->stripe_stage = 1;
    return JEA_OK;
}

int jea_stripe_bits_async(jea_ctx* ctx, const int16_t* d_prev_dc, uint64_t* d_facts) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_facts) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_bits_async: invalid argument");
    if (ctx->stripe_stage != 1) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_bits_async: call jea_stripe_transform_async first");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    EntropyScratch sc;
    int rc;
    if ((rc = prepare_entropy_scratch(ctx, 1, ctx->stripe_mcus, ctx->stripe_capacity, sc, true))) return rc;
    sc.chunk_masks = (const uint8_t*)ctx->masks.ptr;
    launch_stripe_bits(ctx->stream, (const int16_t*)ctx->coefs.ptr, ctx->stripe_mcus, ctx->d_huff, sc, StripeInfo{},
                       (uint32_t*)ctx->small.ptr, ctx->d_status, d_prev_dc, d_facts);
    JEA_HIP(ctx, hipGetLastError());
    ctx->stripe_stage = 2;
    return JEA_OK;
}

int jea_stripe_finish_async(jea_ctx* ctx, const uint64_t* d_all_facts, uint32_t stripe_index, uint32_t n_stripes,
                            uint16_t image_width, uint16_t image_height, const uint16_t lum_q[64],
                            const uint16_t chr_q[64], uint8_t* d_segment, size_t capacity, uint64_t* d_segment_bytes,
                            int32_t* d_status) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_all_facts || !d_segment || !d_segment_bytes || !d_status || capacity == 0 || n_stripes == 0 ||
        stripe_index >= n_stripes || !tables_ok(lum_q, chr_q) || !valid_dims(image_width, image_height))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish_async: invalid argument");
    if (ctx->stripe_stage != 2) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish_async: call jea_stripe_bits_async first");
    const bool is_first = stripe_index == 0, has_next = stripe_index + 1 < n_stripes;
    if (has_next && ctx->stripe_height % 8 != 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stripe_finish_async: only the last stripe may have a height that is not a multiple of 8");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    EntropyScratch sc;
    int rc;
    if ((rc = prepare_entropy_scratch(ctx, 1, ctx->stripe_mcus, ctx->stripe_capacity, sc, true))) return rc;
    StripeInfo* d_st = (StripeInfo*)((uint8_t*)ctx->small.ptr + 256);
    launch_stripe_place(ctx->stream, d_all_facts, stripe_index, n_stripes, d_st);
    HeaderBytes hb{};
    if (is_first) jea_write_header(hb.b, image_width, image_height, lum_q, chr_q);
    launch_stripe_finish(ctx->stream, ctx->stripe_mcus, sc, StripeInfo{}, hb, is_first ? JEA_HEADER_BYTES : 0, !has_next,
                         d_segment, capacity, (uint64_t*)ctx->out_offsets.ptr, d_segment_bytes, ctx->d_status, d_st);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipMemcpyAsync(d_status, ctx->d_status, sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->stripe_stage = 0;
    return JEA_OK;
}

// ---- decode -------------------------------------------------------------------

// Uploads the scan ranges (file ranges minus `skip` leading header bytes), sizes the scratch of the
// parallel decoder, enqueues it followed by `follow_up` (the consumer of the coefficients, if any),
// waits once and reports headers and device status under the name `what`.  Headers travel to the
// host while the decoder runs and are judged first, as the reference does (encoder.cpp:48); what the
// kernels made of a file with a bad header is bounded and thrown away.
static int run_entropy_decode(jea_ctx* ctx, const uint8_t* d_data, const uint64_t* h_offsets, const uint64_t* h_sizes,
                              size_t n_images, uint64_t skip, size_t mcus, int16_t* d_coefs, const char* what,
                              uint32_t* d_block_records = nullptr, const DecodeFollowUp& follow_up = nullptr) {
    int rc;
    const bool with_headers = skip == JEA_HEADER_BYTES;
    uint64_t max_size = 0, total_size = 0;
    bool too_short = false;   // reported after the headers have been looked at, as the reference would
    std::vector<uint64_t> scan_sizes(n_images);
    for (size_t i = 0; i < n_images; ++i) {
        if (h_sizes[i] < skip) return fail(ctx, JEA_ERR_BAD_HEADER, "Failed to find SOI marker");
        if (h_sizes[i] < skip + 2) too_short = true;
        scan_sizes[i] = h_sizes[i] - skip;
        if (scan_sizes[i] > max_size) max_size = scan_sizes[i];
        total_size += scan_sizes[i];
    }
    if (mcus * 3 > 0x7fffffffull || max_size > (1ull << 34))
        return fail(ctx, JEA_ERR_INVALID_ARG, "decode: image or stream too large");
    // Every image gets the workgroups its own scan needs; with many images the kernels look their image up in a
    // table, which travels with the scan ranges.
    const DecodePlan plan = plan_entropy_decode(n_images, scan_sizes.data(), max_size, mcus, total_size);
    if ((uint64_t)plan.total_subs > 0x7fffffffull)
        return fail(ctx, JEA_ERR_INVALID_ARG, "decode: streams too large for one call");
    // host to device: offsets | sizes | unstuffed bases | first workgroup of every image | (image, group) per workgroup
    const size_t seg_bytes = 3 * n_images * sizeof(uint64_t);
    const size_t first_bytes = plan.mapped ? (n_images * sizeof(uint32_t) + 7) / 8 * 8 : 0;
    const size_t map_bytes = plan.mapped ? (size_t)plan.total_groups * 2 * sizeof(uint32_t) : 0;
    const size_t up_bytes = seg_bytes + first_bytes + map_bytes;
    const size_t hdr_bytes = with_headers ? n_images * JEA_HEADER_BYTES : 0;
    if ((rc = ensure_pinned(ctx, 64 + up_bytes + hdr_bytes))) return rc;
    DecodeReport* report = (DecodeReport*)ctx->pinned;
    uint64_t* seg = (uint64_t*)((uint8_t*)ctx->pinned + 64);
    uint8_t* hdr = (uint8_t*)ctx->pinned + 64 + up_bytes;
    uint64_t ubytes = 0;
    for (size_t i = 0; i < n_images; ++i) {
        seg[i] = h_offsets[i] + skip;
        seg[n_images + i] = scan_sizes[i];
        seg[2 * n_images + i] = ubytes;
        ubytes += (scan_sizes[i] + 15) / 16 * 16 + 16;
    }
    if (plan.mapped)
        fill_group_map(plan, scan_sizes.data(), (uint32_t*)((uint8_t*)seg + seg_bytes),
                       (uint32_t*)((uint8_t*)seg + seg_bytes + first_bytes));
    const size_t subs = plan.total_subs, groups = plan.total_groups;
    if ((rc = ensure(ctx, ctx->seg_offsets, up_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->ustream, ubytes))) return rc;
    if ((rc = ensure(ctx, ctx->chunk_drops, n_images * plan.n_chunks * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->chunk_base, n_images * plan.n_chunks * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->img_drops, n_images * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sub_end, subs * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sub_used, subs * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sub_info, subs * sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->grp_sum, groups * sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->grp_base, groups * sizeof(int4)))) return rc;
    if ((rc = ensure(ctx, ctx->round_flags, 4 * sizeof(uint32_t)))) return rc;
    constexpr uint32_t kContextSlots = 1u << 18;
    if ((rc = ensure(ctx, ctx->grp_pending, groups * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->grp_end, groups * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sub_guess, subs * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->first_bad, n_images * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ctx_keys, kContextSlots * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(ctx, ctx->ctx_vals, kContextSlots * sizeof(uint32_t)))) return rc;
    // the pinned block is this context's and the call ends with a wait: the copy may run behind the host;
    // a few images' worth goes as kernel arguments instead, which is quicker than the copy engine
    // files with per-image Huffman tables: the tables are built from the headers first, which needs the scan
    // offsets in device memory before the decoder's first kernel
    const bool own_tables = ctx->optimal_huffman && with_headers;
    if (own_tables && (rc = ensure(ctx, ctx->step_tables, n_images * sizeof(StepTables)))) return rc;
    DecodePrologue prologue;
    if (3 * n_images <= kSeedWords && !own_tables) {
        prologue.h_words = seg;
        prologue.n_words = (uint32_t)(3 * n_images);
    } else {
        JEA_HIP(ctx, hipMemcpyAsync(ctx->seg_offsets.ptr, seg, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    const uint64_t* d_seg = (const uint64_t*)ctx->seg_offsets.ptr;
    auto headers_ok = [&]() -> int {
        for (size_t i = 0; i < n_images; ++i) {
            const char* msg = nullptr;
            if (jea_check_header(hdr + i * JEA_HEADER_BYTES, JEA_HEADER_BYTES, nullptr, nullptr, &msg) != JEA_OK)
                return fail(ctx, JEA_ERR_BAD_HEADER, msg);
        }
        return JEA_OK;
    };
    if (with_headers) {
        // gathered by the decoder's first kernel straight into the pinned block: a copy to the host in this place
        // would hold up the decoder's kernels behind it for the latency of the copy engine (about 20 us)
        prologue.headers_out = hdr;
        prologue.header_bytes = JEA_HEADER_BYTES;
    }
    if (too_short) {   // nothing to decode: only the order of the two messages matters
        if (with_headers) {
            for (size_t i = 0; i < n_images; ++i)
                JEA_HIP(ctx, hipMemcpyAsync(hdr + i * JEA_HEADER_BYTES, d_data + h_offsets[i], JEA_HEADER_BYTES,
                                            hipMemcpyDeviceToHost, ctx->stream));
            JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if ((rc = headers_ok())) return rc;
        }
        return fail(ctx, JEA_ERR_NO_EOI, "Failed to find EOI marker");
    }
    DecodeScratch sc;
    sc.ubase = d_seg + 2 * n_images;
    if (plan.mapped) {
        sc.img_first = (const uint32_t*)((const uint8_t*)ctx->seg_offsets.ptr + seg_bytes);
        sc.grp_map = (const uint32_t*)((const uint8_t*)ctx->seg_offsets.ptr + seg_bytes + first_bytes);
    }
    sc.ustream = (uint8_t*)ctx->ustream.ptr;
    sc.chunk_drops = (uint32_t*)ctx->chunk_drops.ptr;
    sc.chunk_base = (uint64_t*)ctx->chunk_base.ptr;
    sc.img_drops = (uint64_t*)ctx->img_drops.ptr;
    sc.sub_end = (uint32_t*)ctx->sub_end.ptr;
    sc.sub_used = (uint32_t*)ctx->sub_used.ptr;
    sc.sub_info = (int4*)ctx->sub_info.ptr;
    sc.grp_sum = (int4*)ctx->grp_sum.ptr;
    sc.grp_base = (int4*)ctx->grp_base.ptr;
    sc.changed = (uint32_t*)ctx->round_flags.ptr;
    sc.out_records = d_block_records;
    sc.mode = ctx->mode;
    sc.grp_pending = (uint32_t*)ctx->grp_pending.ptr;
    sc.grp_end = (uint32_t*)ctx->grp_end.ptr;
    sc.tables_per_image = own_tables;
    const StepTables* step_tables = ctx->d_steps;
    if (own_tables) {
        launch_build_step_tables(ctx->stream, d_data, d_seg, n_images, JEA_HEADER_BYTES, (StepTables*)ctx->step_tables.ptr);
        JEA_HIP(ctx, hipGetLastError());
        step_tables = (const StepTables*)ctx->step_tables.ptr;
    }
    sc.sub_guess = (uint32_t*)ctx->sub_guess.ptr;
    sc.first_bad = (uint32_t*)ctx->first_bad.ptr;
    sc.ctx_keys = (unsigned long long*)ctx->ctx_keys.ptr;
    sc.ctx_vals = (uint32_t*)ctx->ctx_vals.ptr;
    sc.ctx_slots = kContextSlots;
    ctx->n_marks = 0;
    if (ctx->profiling) {
        JEA_HIP(ctx, hipEventRecord(ctx->marks[0], ctx->stream));
        sc.marks = ctx->marks + 1;
    }
    // First without an unstuffed copy (the kernels read the raw scans).  A stream that two synchronisation rounds
    // do not settle -- exactly periodic ones, mostly -- goes through the variant with an unstuffing pass, which
    // has the repair rounds.
    JEA_HIP(ctx, launch_entropy_decode(ctx->stream, d_data, d_seg, d_seg + n_images, plan, sc, step_tables, d_coefs,
                                       ctx->d_status, follow_up, report, prologue, true));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling) ctx->n_marks = 5;   // unstuffing | synchronisation rounds | scan and write pass | follow-up
    if (with_headers && (rc = headers_ok())) return rc;
    if (report->unsettled != 0) {
        // what produces the result from here on is not the pass whose phases were marked: no phase times for this call
        sc.marks = nullptr;
        ctx->n_marks = 0;
        JEA_HIP(ctx, launch_entropy_decode(ctx->stream, d_data, d_seg, d_seg + n_images, plan, sc, step_tables, d_coefs,
                                           ctx->d_status, follow_up, report, prologue, false));
        JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (report->unsettled != 0)
            JEA_HIP(ctx, repair_entropy_decode(ctx->stream, d_seg + n_images, plan, sc, step_tables, d_coefs, ctx->d_status,
                                               follow_up, report));
    }
    return report_status(ctx, report->status, what);
}

int jea_remove_stuffing(jea_ctx* ctx, const uint8_t* d_scan, size_t len, uint8_t* d_out, size_t* out_len) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_scan || !d_out || !out_len || len == 0 || len > ((size_t)1 << 34))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_remove_stuffing: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    const DecodePlan plan = plan_entropy_decode(1, nullptr, len, 1);
    int rc;
    if ((rc = ensure(ctx, ctx->seg_offsets, 3 * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->ustream, (len + 15) / 16 * 16 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->chunk_drops, plan.n_chunks * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->chunk_base, plan.n_chunks * sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->img_drops, sizeof(uint64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->round_flags, 4 * sizeof(uint32_t)))) return rc;
    const uint64_t seg[3] = {0, (uint64_t)len, 0};   // offset | size | base of the unstuffed copy
    DecodePrologue prologue;
    prologue.h_words = seg;
    prologue.n_words = 3;
    const uint64_t* d_seg = (const uint64_t*)ctx->seg_offsets.ptr;
    DecodeScratch sc{};
    sc.ubase = d_seg + 2;
    sc.ustream = (uint8_t*)ctx->ustream.ptr;
    sc.chunk_drops = (uint32_t*)ctx->chunk_drops.ptr;
    sc.chunk_base = (uint64_t*)ctx->chunk_base.ptr;
    sc.img_drops = (uint64_t*)ctx->img_drops.ptr;
    sc.changed = (uint32_t*)ctx->round_flags.ptr;
    JEA_HIP(ctx, launch_unstuff(ctx->stream, d_scan, d_seg, d_seg + 1, plan, sc, ctx->d_status, prologue));
    uint64_t drops = 0;
    JEA_HIP(ctx, hipMemcpyAsync(&drops, sc.img_drops, sizeof(drops), hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(d_out, sc.ustream, len - drops, hipMemcpyDeviceToDevice, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out_len = len - (size_t)drops;
    return JEA_OK;
}

int jea_entropy_decode(jea_ctx* ctx, const uint8_t* d_scan, const uint64_t* h_offsets, const uint64_t* h_sizes,
                       size_t n_images, size_t mcus_per_image, int16_t* d_coefs) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_scan || !h_offsets || !h_sizes || !d_coefs || n_images == 0 || mcus_per_image == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_entropy_decode: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    if (n_images > 65535) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_entropy_decode: more than 65535 images");
    return run_entropy_decode(ctx, d_scan, h_offsets, h_sizes, n_images, 0, mcus_per_image, d_coefs,
                              "jea_entropy_decode");
}

int jea_decode_batch(jea_ctx* ctx, const uint8_t* d_jpeg, const uint64_t* h_offsets, const uint64_t* h_sizes,
                     size_t n_images, uint16_t width, uint16_t height, const uint16_t lum_q[64],
                     const uint16_t chr_q[64], uint8_t* d_rgb) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!d_jpeg || !h_offsets || !h_sizes || !d_rgb || n_images == 0 || n_images > 65535 ||
        !valid_dims(width, height) || !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_decode_batch: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < n_images; ++i)
        if (h_sizes[i] < JEA_HEADER_BYTES) return fail(ctx, JEA_ERR_BAD_HEADER, "Failed to find SOI marker");
    const size_t mcus = mcus_of(width, height);
    int rc;
    if ((rc = ensure(ctx, ctx->coefs, n_images * mcus * 3 * 64 * sizeof(int16_t)))) return rc;
    // the coefficients stay inside the library: block records and chunk planes
    if ((rc = ensure(ctx, ctx->masks, n_images * mcus * 3 * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->group_flags, idct_group_list_bytes()))) return rc;
    // The inverse transform is enqueued right behind the write pass, before anything is known about the
    // stream: a decode waits for the GPU once.  So when an error is reported, d_rgb holds garbage.
    const QuantTables qt = to_float_tables(lum_q, chr_q);
    // Which inverse-transform route suits the batch is read from the files' sizes.  An MCU of one colour takes
    // 1.75 to about 2.2 bytes of the stream (three short DC codes, three end-of-block codes), one with detail more.
    // Measured on batches of 256 smooth 1080p frames (the two-kernel route against the general kernel, us):
    // q=25 1.83 B/MCU 352 / 447, q=50 1.90 B 501 / 593, q=60 1.99 B 560 / 703 -- but q=75 2.74 B 1812 / 1026 and
    // q=85 4.99 B 1894 / 1296, where most groups of 64 MCUs hold more than 16 MCUs with detail.  An average says
    // little (a few noise frames among smooth ones lift it past 2.7): nine files in ten must be under 2.2 B/MCU.
    // (With the pass-through mapper all three components use the luminance codes, whose end-of-block code is two
    // bits longer than the chrominance one: 2.25 to about 2.8 bytes for an MCU of one colour.)
#ifdef JEA_SMALL_LIMIT   // experiments
    const double small_limit = JEA_SMALL_LIMIT;
#else
    const double small_limit = ctx->mode.passthrough ? 2.8 : 2.2;
#endif
    size_t small_files = 0;
    for (size_t i = 0; i < n_images; ++i)
        small_files += (double)(h_sizes[i] - JEA_HEADER_BYTES) < small_limit * (double)mcus;
    // (files with per-image Huffman tables are smaller for the same content: their sizes say nothing here)
    const bool mostly_single_colours = !ctx->optimal_huffman && small_files * 10 >= n_images * 9;
    auto inverse = [&](const DecodeReportRelay& relay) {
        launch_idct_dequant(ctx->stream, (const int16_t*)ctx->coefs.ptr, n_images, width, height, qt, d_rgb,
                            (const uint32_t*)ctx->masks.ptr, ctx->mode, relay, (uint32_t*)ctx->group_flags.ptr,
                            mostly_single_colours);
        return true;
    };
    return run_entropy_decode(ctx, d_jpeg, h_offsets, h_sizes, n_images, JEA_HEADER_BYTES, mcus,
                              (int16_t*)ctx->coefs.ptr, "jea_decode_batch", (uint32_t*)ctx->masks.ptr, inverse);
}

int jea_decode_host(jea_ctx* ctx, const uint8_t* h_jpeg, size_t len, uint16_t width, uint16_t height,
                    const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* h_rgb) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_jpeg || !h_rgb || !valid_dims(width, height))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_decode_host: invalid argument");
    const char* msg = nullptr;
    if (jea_check_header(h_jpeg, len, nullptr, nullptr, &msg) != JEA_OK) return fail(ctx, JEA_ERR_BAD_HEADER, msg);
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    const size_t out_bytes = (size_t)3 * width * height;
    int rc;
    if ((rc = ensure(ctx, ctx->stage_out, len))) return rc;
    if ((rc = ensure(ctx, ctx->stage_in, out_bytes))) return rc;
    JEA_HIP(ctx, hipMemcpyAsync(ctx->stage_out.ptr, h_jpeg, len, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t off = 0, size = len;
    rc = jea_decode_batch(ctx, (const uint8_t*)ctx->stage_out.ptr, &off, &size, 1, width, height, lum_q, chr_q,
                          (uint8_t*)ctx->stage_in.ptr);
    if (rc) return rc;
    return download_pipelined(ctx, h_rgb, ctx->stage_in.ptr, out_bytes);
}

// ---- batches in host memory ----------------------------------------------------
//
// The caller of the reference holds bitmaps and files in host memory (examples/encode-decode.cpp:104-118).  A batch
// of them goes through the GPU in pieces of about 32 MB of pixels on two staging slots, so that three things overlap:
// the host copies the next piece into pinned memory and the copy engine moves it over the link (one stream), the
// kernels work on the current piece (the context's stream), and the piece before travels back (another stream)
// and is copied to the caller's memory.  Caller memory that is pinned already (jea_host_malloc) skips the host copies.
// The streams are ordered by events that are always recorded before they are waited for, all at one priority.

namespace {

int ensure_host(jea_ctx* ctx, void*& ptr, size_t& have, size_t want) {
    if (want <= have) return JEA_OK;
    if (ptr) JEA_HIP(ctx, hipHostFree(ptr));
    ptr = nullptr;
    have = 0;
    const size_t bytes = want + want / 8 + 4096;
    JEA_HIP(ctx, hipHostMalloc(&ptr, bytes, hipHostMallocDefault));
    have = bytes;
    return JEA_OK;
}

int prepare_pipe(jea_ctx* ctx) {
    if (!ctx->copy_in) JEA_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
    if (!ctx->copy_out) JEA_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
    for (PipeSlot& sl : ctx->pipe)
        for (hipEvent_t* e : {&sl.h2d, &sl.work, &sl.d2h})
            if (!*e) JEA_HIP(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    return JEA_OK;
}

// Everything the pipeline has in flight, before its buffers change size or the call returns.
int drain_pipe(jea_ctx* ctx) {
    JEA_HIP(ctx, hipStreamSynchronize(ctx->copy_in));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->copy_out));
    return JEA_OK;
}

// Memory the GPU can reach directly (hipHostMalloc, jea_host_malloc)?
bool is_pinned(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Pixels on their way back to host memory that the GPU can write (pinned): a kernel of a few workgroups instead of
// hipMemcpyAsync.  The runtime's copy from device memory to pinned host memory is a kernel too, one that fills the
// GPU: the decoder's kernels of the next piece, which should run beside it, got their workgroups in only as its
// workgroups left (a 4K picture's first decode kernel then took 180 us instead of 34), the next copy started late,
// and the link idled a fifth of the time.  JEA_TO_HOST_GROUPS workgroups keep the link as busy and leave the rest of
// the GPU alone.
#ifndef JEA_TO_HOST_GROUPS
#define JEA_TO_HOST_GROUPS 32
#endif
typedef uint32_t Words4 __attribute__((ext_vector_type(4)));   // a built-in vector type, as the non-temporal loads want
__global__ __launch_bounds__(256) void to_host_kernel(const Words4* __restrict__ src, Words4* __restrict__ dst, size_t n16,
                                                      size_t tail_bytes) {
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // four loads, then four stores: enough bytes under way per lane to cover the link's latency
    for (; i + 3 * step < n16; i += 4 * step) {
        const Words4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + step),
                     c = __builtin_nontemporal_load(src + i + 2 * step), d = __builtin_nontemporal_load(src + i + 3 * step);
        dst[i] = a;
        dst[i + step] = b;
        dst[i + 2 * step] = c;
        dst[i + 3 * step] = d;
    }
    for (; i < n16; i += step) dst[i] = __builtin_nontemporal_load(src + i);
    if (blockIdx.x == 0 && threadIdx.x < tail_bytes)
        reinterpret_cast<uint8_t*>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint8_t*>(src + n16)[threadIdx.x];
}

// Device memory to pinned host memory on `stream`: the kernel above where the addresses allow 16-byte pieces.
// `chunk`: bytes per launch.  A kernel of another stream that runs beside such a copy does not end before the copy's
// launch has ended (the first decode kernel of the next piece, 34 us alone, ended 20 us after a whole-piece copy of
// 455 us, whatever it wrote), so a caller with time to spare for the launches cuts the copy into short ones:
// 64 x 4K pictures into the caller's pinned memory, one launch per piece 46.3 GB/s, 2 MB per launch 48.5, 1 MB 49.3;
// into staging memory that the host then copies on (pageable caller memory: the host is the busy one) 43.5 / 42.2 / 40.9.
int copy_to_pinned(jea_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, hipStream_t stream, size_t chunk) {
    void* mapped = nullptr;
    if (bytes >= (1u << 20) && (reinterpret_cast<uintptr_t>(d_src) & 15u) == 0 &&
        hipHostGetDevicePointer(&mapped, h_dst, 0) == hipSuccess && mapped != nullptr &&
        (reinterpret_cast<uintptr_t>(mapped) & 15u) == 0) {
        chunk = (chunk + 15) / 16 * 16;
        for (size_t at = 0; at < bytes; at += chunk) {
            const size_t len = bytes - at < chunk ? bytes - at : chunk;
            to_host_kernel<<<JEA_TO_HOST_GROUPS, 256, 0, stream>>>((const Words4*)((const uint8_t*)d_src + at),
                                                                   (Words4*)((uint8_t*)mapped + at), len / 16, len % 16);
        }
        JEA_HIP(ctx, hipGetLastError());
        return JEA_OK;
    }
    (void)hipGetLastError();
    JEA_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
    return JEA_OK;
}

size_t images_per_piece(size_t image_bytes, size_t n_images) {
    size_t k = ((size_t)32 << 20) / image_bytes;
    if (k < 1) k = 1;
    if (k > n_images) k = n_images;
    if (k > 4096) k = 4096;
    return k;
}

}  // namespace

int jea_host_malloc(jea_ctx* ctx, size_t bytes, void** h_ptr) {
    if (!ctx || !h_ptr || bytes == 0) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return JEA_OK;
}

int jea_host_free(jea_ctx* ctx, void* h_ptr) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipHostFree(h_ptr));
    return JEA_OK;
}

int jea_encode_host_batch(jea_ctx* ctx, const uint8_t* h_rgb, size_t n_images, uint16_t width, uint16_t height,
                          const uint16_t lum_q[64], const uint16_t chr_q[64], uint8_t* h_out, size_t out_capacity,
                          uint64_t* h_offsets, uint64_t* h_sizes) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_rgb || !h_out || !h_offsets || !h_sizes || n_images == 0 || !valid_dims(width, height) ||
        !tables_ok(lum_q, chr_q) || out_capacity < JEA_HEADER_BYTES + 3)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_encode_host_batch: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = prepare_pipe(ctx))) return rc;
    const size_t image_bytes = (size_t)3 * width * height;
    const size_t k = images_per_piece(image_bytes, n_images);
    const size_t n_pieces = (n_images + k - 1) / k;
    const size_t worst = k * ((jea_max_jpeg_bytes(width, height) + 63) / 64 * 64);
    const size_t piece_cap = worst < out_capacity ? worst : out_capacity;   // no piece can need more than the caller has room for
    const size_t meta_bytes = (2 * k + 1) * sizeof(uint64_t);               // offsets | sizes | status
    const bool in_pinned = is_pinned(h_rgb), out_pinned = is_pinned(h_out);
    if ((rc = drain_pipe(ctx))) return rc;
    for (PipeSlot& sl : ctx->pipe) {
        if ((rc = ensure(ctx, sl.d_in, k * image_bytes))) return rc;
        if ((rc = ensure(ctx, sl.d_out, piece_cap))) return rc;
        if ((rc = ensure(ctx, sl.d_meta, meta_bytes))) return rc;
        if (!in_pinned && (rc = ensure_host(ctx, sl.pin_in, sl.pin_in_bytes, k * image_bytes))) return rc;
        if ((rc = ensure_host(ctx, sl.pin_meta, sl.pin_meta_bytes, meta_bytes))) return rc;
    }
    // (the context's own scratch grows inside the first piece's encode call, before anything else is in flight)
    auto images_of = [&](size_t p) { return p + 1 < n_pieces ? k : n_images - p * k; };
    uint64_t out_at = 0;      // where the next piece's files start in h_out (a multiple of 64)
    struct Placed { uint64_t at, bytes; };
    Placed placed[2] = {{0, 0}, {0, 0}};
    int result = JEA_OK;
    // three stages, one piece apart: [A] piece p goes in and is encoded, [B] the sizes of piece p-1 arrive and its
    // files start back, [C] the files of piece p-2 reach the caller's memory
    for (size_t p = 0; p < n_pieces + 2 && result == JEA_OK; ++p) {
        if (p < n_pieces) {
            PipeSlot& sl = ctx->pipe[p & 1];
            const size_t n = images_of(p), bytes = n * image_bytes;
            const uint8_t* src = h_rgb + p * k * image_bytes;
            // the slot's staging memory is free once the copy of piece p-2 has left it, its device input once
            // that piece has been encoded
            JEA_HIP(ctx, hipEventSynchronize(sl.h2d));
            JEA_HIP(ctx, hipStreamWaitEvent(ctx->copy_in, sl.work, 0));
            if (!in_pinned) {
                parallel_memcpy(sl.pin_in, src, bytes);
                src = (const uint8_t*)sl.pin_in;
            }
            JEA_HIP(ctx, hipMemcpyAsync(sl.d_in.ptr, src, bytes, hipMemcpyHostToDevice, ctx->copy_in));
            JEA_HIP(ctx, hipEventRecord(sl.h2d, ctx->copy_in));
            JEA_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.h2d, 0));
            JEA_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.d2h, 0));   // the files of piece p-2 have left d_out
            uint64_t* m = (uint64_t*)sl.d_meta.ptr;
            rc = jea_encode_batch_async(ctx, (const uint8_t*)sl.d_in.ptr, n, width, height, lum_q, chr_q,
                                        (uint8_t*)sl.d_out.ptr, piece_cap, m, m + k, (int32_t*)(m + 2 * k));
            if (rc) { result = rc; break; }
            JEA_HIP(ctx, hipMemcpyAsync(sl.pin_meta, sl.d_meta.ptr, meta_bytes, hipMemcpyDeviceToHost, ctx->stream));
            JEA_HIP(ctx, hipEventRecord(sl.work, ctx->stream));
        }
        if (p >= 1 && p - 1 < n_pieces) {
            const size_t q = p - 1;
            PipeSlot& sl = ctx->pipe[q & 1];
            const size_t n = images_of(q);
            JEA_HIP(ctx, hipEventSynchronize(sl.work));
            const uint64_t* m = (const uint64_t*)sl.pin_meta;
            const int32_t st = (int32_t)m[2 * k];
            if (st != 0) { result = report_status(ctx, st, "jea_encode_host_batch"); break; }
            const uint64_t bytes = m[n - 1] + m[k + n - 1];       // files lie back to back from offset 0 at 64-byte steps
            if (out_at + bytes > out_capacity) {
                result = fail(ctx, JEA_ERR_CAPACITY, "jea_encode_host_batch: output capacity too small");
                break;
            }
            for (size_t i = 0; i < n; ++i) {
                h_offsets[q * k + i] = out_at + m[i];
                h_sizes[q * k + i] = m[k + i];
            }
            uint8_t* dst = h_out + out_at;
            if (!out_pinned) {
                // the slot's pinned output is free: piece q-2 was copied out of it in stage C of the round before
                if ((rc = ensure_host(ctx, sl.pin_out, sl.pin_out_bytes, bytes))) { result = rc; break; }
                dst = (uint8_t*)sl.pin_out;
            }
            if ((rc = copy_to_pinned(ctx, dst, sl.d_out.ptr, bytes, ctx->copy_out, bytes))) { result = rc; break; }
            JEA_HIP(ctx, hipEventRecord(sl.d2h, ctx->copy_out));
            placed[q & 1] = Placed{out_at, bytes};
            out_at = (out_at + bytes + 63) / 64 * 64;
        }
        if (p >= 2) {
            PipeSlot& sl = ctx->pipe[p & 1];
            JEA_HIP(ctx, hipEventSynchronize(sl.d2h));
            if (!out_pinned) parallel_memcpy(h_out + placed[p & 1].at, sl.pin_out, placed[p & 1].bytes);
        }
    }
    const int drained = drain_pipe(ctx);
    return result != JEA_OK ? result : drained;
}

int jea_decode_host_batch(jea_ctx* ctx, const uint8_t* h_jpeg, const uint64_t* h_offsets, const uint64_t* h_sizes,
                          size_t n_images, uint16_t width, uint16_t height, const uint16_t lum_q[64],
                          const uint16_t chr_q[64], uint8_t* h_rgb) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_jpeg || !h_offsets || !h_sizes || !h_rgb || n_images == 0 || !valid_dims(width, height) || !tables_ok(lum_q, chr_q))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_decode_host_batch: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = prepare_pipe(ctx))) return rc;
    // the headers are judged first, as the reference would (encoder.cpp:48)
    for (size_t i = 0; i < n_images; ++i) {
        const char* msg = nullptr;
        if (jea_check_header(h_jpeg + h_offsets[i], (size_t)h_sizes[i], nullptr, nullptr, &msg) != JEA_OK)
            return fail(ctx, JEA_ERR_BAD_HEADER, msg);
    }
    const size_t image_bytes = (size_t)3 * width * height;
    const size_t k = images_per_piece(image_bytes, n_images);
    const size_t n_pieces = (n_images + k - 1) / k;
    const bool out_pinned = is_pinned(h_rgb);
    auto images_of = [&](size_t p) { return p + 1 < n_pieces ? k : n_images - p * k; };
    size_t max_files = 0;   // bytes of the files of the largest piece, each rounded up to 64
    for (size_t p = 0; p < n_pieces; ++p) {
        size_t total = 0;
        for (size_t i = 0; i < images_of(p); ++i) total += ((size_t)h_sizes[p * k + i] + 63) / 64 * 64;
        if (total > max_files) max_files = total;
    }
    if ((rc = drain_pipe(ctx))) return rc;
    for (PipeSlot& sl : ctx->pipe) {
        if ((rc = ensure(ctx, sl.d_in, max_files))) return rc;
        if ((rc = ensure(ctx, sl.d_out, k * image_bytes))) return rc;
        if ((rc = ensure_host(ctx, sl.pin_in, sl.pin_in_bytes, max_files))) return rc;
        if (!out_pinned && (rc = ensure_host(ctx, sl.pin_out, sl.pin_out_bytes, k * image_bytes))) return rc;
    }
    std::vector<uint64_t> offs(k), sizes(k);
    int result = JEA_OK;
    // per round: the pixels of piece p-2 reach the caller's memory, then piece p's files go in and are decoded (the
    // decode waits for the GPU itself), then its pixels start back -- while the pixels of piece p-1 are still on the link
    // (pageable caller memory: that last copy, a third of a millisecond of host time for a piece, runs on a thread of
    // its own beside the decode call of the round and is waited for before the slot's staging memory is written again)
    std::thread to_caller;
    auto join_to_caller = [&to_caller] { if (to_caller.joinable()) to_caller.join(); };
    for (size_t p = 0; p < n_pieces + 2 && result == JEA_OK; ++p) {
        if (p >= 2) {
            PipeSlot& sl = ctx->pipe[p & 1];
            const hipError_t e = hipEventSynchronize(sl.d2h);
            if (e != hipSuccess) { result = fail(ctx, JEA_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e)); break; }
            if (!out_pinned) {
                uint8_t* const to = h_rgb + (p - 2) * k * image_bytes;
                const void* const from = sl.pin_out;
                const size_t bytes = images_of(p - 2) * image_bytes;
                to_caller = std::thread([=] { parallel_memcpy(to, from, bytes); });
            }
        }
        if (p < n_pieces) {
            PipeSlot& sl = ctx->pipe[p & 1];
            const size_t n = images_of(p);
            size_t at = 0;
            for (size_t i = 0; i < n; ++i) {
                offs[i] = at;
                sizes[i] = h_sizes[p * k + i];
                std::memcpy((uint8_t*)sl.pin_in + at, h_jpeg + h_offsets[p * k + i], (size_t)sizes[i]);
                at += ((size_t)sizes[i] + 63) / 64 * 64;
            }
            {   // (no early return in this loop: the thread above may be running)
                const hipError_t e = hipMemcpyAsync(sl.d_in.ptr, sl.pin_in, at, hipMemcpyHostToDevice, ctx->stream);
                if (e != hipSuccess) { result = fail(ctx, JEA_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e)); break; }
            }
            rc = jea_decode_batch(ctx, (const uint8_t*)sl.d_in.ptr, offs.data(), sizes.data(), n, width, height, lum_q, chr_q,
                                  (uint8_t*)sl.d_out.ptr);
            if (rc) { result = rc; break; }
            join_to_caller();   // piece p-2 has left this slot's staging memory
            uint8_t* dst = out_pinned ? h_rgb + p * k * image_bytes : (uint8_t*)sl.pin_out;
            if ((rc = copy_to_pinned(ctx, dst, sl.d_out.ptr, n * image_bytes, ctx->copy_out,
                                     out_pinned ? (size_t)1 << 20 : n * image_bytes))) { result = rc; break; }
            const hipError_t e = hipEventRecord(sl.d2h, ctx->copy_out);
            if (e != hipSuccess) { result = fail(ctx, JEA_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e)); break; }
        }
        if (p >= n_pieces) join_to_caller();
    }
    join_to_caller();
    const int drained = drain_pipe(ctx);
    return result != JEA_OK ? result : drained;
}

// ---- component-level entry points ---------------------------------------------

int jea_rgb_to_ycbcr_host(jea_ctx* ctx, const uint8_t* h_rgb, size_t n_pixels, uint8_t* h_ycc) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_rgb || !h_ycc || n_pixels == 0) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_rgb_to_ycbcr_host: invalid argument");
    return staged_call(ctx, h_rgb, 3 * n_pixels, h_ycc, 3 * n_pixels, [&](void* in, void* out) {
        launch_rgb_to_ycbcr(ctx->stream, (const uint8_t*)in, n_pixels, (uint8_t*)out);
    });
}

int jea_ycbcr_to_rgb_host(jea_ctx* ctx, const uint8_t* h_ycc, size_t n_pixels, uint8_t* h_rgb) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_rgb || !h_ycc || n_pixels == 0) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_ycbcr_to_rgb_host: invalid argument");
    return staged_call(ctx, h_ycc, 3 * n_pixels, h_rgb, 3 * n_pixels, [&](void* in, void* out) {
        launch_ycbcr_to_rgb(ctx->stream, (const uint8_t*)in, n_pixels, (uint8_t*)out);
    });
}

int jea_dct_forward_blocks(jea_ctx* ctx, const uint8_t* h_samples, size_t n_blocks, float* h_dct) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_samples || !h_dct || n_blocks == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_dct_forward_blocks: invalid argument");
    return staged_call(ctx, h_samples, 64 * n_blocks, h_dct, 256 * n_blocks, [&](void* in, void* out) {
        launch_dct_forward_blocks(ctx->stream, (const uint8_t*)in, n_blocks, (float*)out);
    });
}

int jea_dct_inverse_blocks(jea_ctx* ctx, const float* h_dct, size_t n_blocks, uint8_t* h_samples) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_samples || !h_dct || n_blocks == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_dct_inverse_blocks: invalid argument");
    return staged_call(ctx, h_dct, 256 * n_blocks, h_samples, 64 * n_blocks, [&](void* in, void* out) {
        launch_dct_inverse_blocks(ctx->stream, (const float*)in, n_blocks, (uint8_t*)out);
    });
}

int jea_quantise_blocks(jea_ctx* ctx, const float* h_dct, size_t n_blocks, const uint16_t table[64],
                        int16_t* h_quantised) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_dct || !h_quantised || n_blocks == 0 || !table_ok(table))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_quantise_blocks: invalid argument");
    const QuantTables q = to_float_tables(table, table);
    return staged_call(ctx, h_dct, 256 * n_blocks, h_quantised, 128 * n_blocks, [&](void* in, void* out) {
        launch_quantise_blocks(ctx->stream, (const float*)in, n_blocks, q, true, (int16_t*)out);
    });
}

int jea_dequantise_blocks(jea_ctx* ctx, const int16_t* h_quantised, size_t n_blocks, const uint16_t table[64],
                          float* h_dct) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_dct || !h_quantised || n_blocks == 0 || !table_ok(table))
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_dequantise_blocks: invalid argument");
    const QuantTables q = to_float_tables(table, table);
    return staged_call(ctx, h_quantised, 128 * n_blocks, h_dct, 256 * n_blocks, [&](void* in, void* out) {
        launch_dequantise_blocks(ctx->stream, (const int16_t*)in, n_blocks, q, true, (float*)out);
    });
}

int jea_huffman_encode_blocks(jea_ctx* ctx, const int16_t* h_zigzag, const int16_t* h_prev_dc,
                              const uint8_t* h_is_luminance, size_t n_blocks, uint16_t* h_lengths, uint32_t* h_bits) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_zigzag || !h_prev_dc || !h_is_luminance || !h_lengths || !h_bits || n_blocks == 0)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_huffman_encode_blocks: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    // input: [zz 128 B][prev_dc 2 B][class 1 B] per block; output: [bits 208 B][length 2 B] per block
    const size_t in_zz = 128 * n_blocks, in_dc = (2 * n_blocks + 15) & ~size_t(15), in_cls = n_blocks;
    const size_t out_bits = (size_t)kSlotWords * 4 * n_blocks, out_len = 2 * n_blocks;
    int rc;
    if ((rc = ensure(ctx, ctx->stage_in, in_zz + in_dc + in_cls))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, out_bits + out_len))) return rc;
    uint8_t* din = (uint8_t*)ctx->stage_in.ptr;
    uint8_t* dout = (uint8_t*)ctx->stage_out.ptr;
    JEA_HIP(ctx, hipMemcpyAsync(din, h_zigzag, in_zz, hipMemcpyHostToDevice, ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(din + in_zz, h_prev_dc, 2 * n_blocks, hipMemcpyHostToDevice, ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(din + in_zz + in_dc, h_is_luminance, n_blocks, hipMemcpyHostToDevice, ctx->stream));
    JEA_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
    JEA_HIP(ctx, hipMemsetAsync(dout, 0, out_bits, ctx->stream));
    launch_huffman_encode_blocks(ctx->stream, (const int16_t*)din, (const int16_t*)(din + in_zz), din + in_zz + in_dc,
                                 n_blocks, ctx->d_huff, (uint16_t*)(dout + out_bits), (uint32_t*)dout, ctx->d_status);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipMemcpyAsync(h_bits, dout, out_bits, hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(h_lengths, dout + out_bits, out_len, hipMemcpyDeviceToHost, ctx->stream));
    return finish_status(ctx, ctx->d_status, "jea_huffman_encode_blocks");
}

int jea_stream_upload(jea_ctx* ctx, const uint8_t* h_stream, size_t len) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!h_stream || len == 0) return fail(ctx, JEA_ERR_INVALID_ARG, "jea_stream_upload: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->stream_copy, len))) return rc;
    JEA_HIP(ctx, hipMemcpyAsync(ctx->stream_copy.ptr, h_stream, len, hipMemcpyHostToDevice, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream_len = len;
    return JEA_OK;
}

int jea_huffman_decode_block(jea_ctx* ctx, uint64_t start_bit, int is_luminance, int16_t* last_dc,
                             int16_t h_zigzag[64], uint64_t* end_bit) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!last_dc || !h_zigzag || !end_bit || ctx->stream_len == 0 || start_bit >= 8 * (uint64_t)ctx->stream_len)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_huffman_decode_block: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->small, 256))) return rc;
    uint8_t* d = (uint8_t*)ctx->small.ptr;  // [zz 128][end_bit 8][last_dc 4]
    JEA_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
    launch_huffman_decode_block(ctx->stream, (const uint8_t*)ctx->stream_copy.ptr, ctx->stream_len, start_bit,
                                is_luminance != 0, (int)*last_dc, ctx->d_decode, (int16_t*)d, (uint64_t*)(d + 128),
                                (int32_t*)(d + 136), ctx->d_status);
    JEA_HIP(ctx, hipGetLastError());
    uint8_t h[144];
    JEA_HIP(ctx, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = finish_status(ctx, ctx->d_status, "jea_huffman_decode_block"))) return rc;
    std::memcpy(h_zigzag, h, 128);
    std::memcpy(end_bit, h + 128, 8);
    int32_t dc;
    std::memcpy(&dc, h + 136, 4);
    *last_dc = (int16_t)dc;
    return JEA_OK;
}

int jea_huffman_encode_rle(jea_ctx* ctx, int16_t dc_difference, const uint16_t* h_runs, const int16_t* h_values,
                           size_t n_entries, int is_luminance, uint32_t* length_bits, uint32_t* h_bits,
                           size_t bits_capacity_words) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    const size_t words = (22 + 26 * n_entries + 31) / 32;
    if (!length_bits || !h_bits || (n_entries && (!h_runs || !h_values)) || n_entries > 4096 || bits_capacity_words < words)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_huffman_encode_rle: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    // input: [runs 2n][values 2n]; output: [bits 4*words][length 4]
    const size_t in_half = (2 * n_entries + 15) & ~size_t(15);
    int rc;
    if ((rc = ensure(ctx, ctx->stage_in, 2 * in_half + 16))) return rc;
    if ((rc = ensure(ctx, ctx->stage_out, 4 * words + 16))) return rc;
    uint8_t* din = (uint8_t*)ctx->stage_in.ptr;
    uint8_t* dout = (uint8_t*)ctx->stage_out.ptr;
    if (n_entries) {
        JEA_HIP(ctx, hipMemcpyAsync(din, h_runs, 2 * n_entries, hipMemcpyHostToDevice, ctx->stream));
        JEA_HIP(ctx, hipMemcpyAsync(din + in_half, h_values, 2 * n_entries, hipMemcpyHostToDevice, ctx->stream));
    }
    JEA_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
    JEA_HIP(ctx, hipMemsetAsync(dout, 0, 4 * words + 4, ctx->stream));
    launch_huffman_encode_rle(ctx->stream, (int)dc_difference, (const uint16_t*)din, (const int16_t*)(din + in_half),
                              (uint32_t)n_entries, is_luminance != 0, ctx->d_huff, (uint32_t*)dout, (uint32_t*)(dout + 4 * words),
                              ctx->d_status);
    JEA_HIP(ctx, hipGetLastError());
    JEA_HIP(ctx, hipMemcpyAsync(h_bits, dout, 4 * words, hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipMemcpyAsync(length_bits, dout + 4 * words, 4, hipMemcpyDeviceToHost, ctx->stream));
    return finish_status(ctx, ctx->d_status, "jea_huffman_encode_rle");
}

int jea_huffman_decode_rle(jea_ctx* ctx, uint64_t start_bit, int is_luminance, int16_t* dc_difference,
                           uint16_t h_runs[64], int16_t h_values[64], size_t* n_entries, uint64_t* end_bit) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    if (!dc_difference || !h_runs || !h_values || !n_entries || !end_bit || ctx->stream_len == 0 ||
        start_bit >= 8 * (uint64_t)ctx->stream_len)
        return fail(ctx, JEA_ERR_INVALID_ARG, "jea_huffman_decode_rle: invalid argument");
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->small, 512))) return rc;
    uint8_t* d = (uint8_t*)ctx->small.ptr;   // [runs 128][values 128][end_bit 8][dc 4][n 4]
    JEA_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int32_t), ctx->stream));
    launch_huffman_decode_rle(ctx->stream, (const uint8_t*)ctx->stream_copy.ptr, ctx->stream_len, start_bit, is_luminance != 0,
                              ctx->d_decode, (int32_t*)(d + 264), (uint16_t*)d, (int16_t*)(d + 128), (uint32_t*)(d + 268),
                              (uint64_t*)(d + 256), ctx->d_status);
    JEA_HIP(ctx, hipGetLastError());
    uint8_t h[272];
    JEA_HIP(ctx, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = finish_status(ctx, ctx->d_status, "jea_huffman_decode_rle"))) return rc;
    uint32_t n;
    int32_t dc;
    std::memcpy(&n, h + 268, 4);
    std::memcpy(&dc, h + 264, 4);
    std::memcpy(end_bit, h + 256, 8);
    std::memcpy(h_runs, h, 2 * (size_t)n);
    std::memcpy(h_values, h + 128, 2 * (size_t)n);
    *n_entries = n;
    *dc_difference = (int16_t)dc;
    return JEA_OK;
}

// ---- device memory helpers ---------------------------------------------------

int jea_malloc(jea_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr || bytes == 0) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipMalloc(d_ptr, bytes));
    return JEA_OK;
}
int jea_free(jea_ctx* ctx, void* d_ptr) {
    if (!ctx) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    JEA_HIP(ctx, hipFree(d_ptr));
    return JEA_OK;
}
int jea_memcpy_h2d(jea_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || !d_dst || !h_src) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}
int jea_memcpy_d2h(jea_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || !h_dst || !d_src) return JEA_ERR_INVALID_ARG;
    JEA_HIP(ctx, hipSetDevice(ctx->device));
    JEA_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JEA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JEA_OK;
}

}  // extern "C"
