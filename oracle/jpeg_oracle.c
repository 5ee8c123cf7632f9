/*
 * jpeg_oracle.c -- CPU restatement of the reference baseline-JPEG encode/decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * library: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; nothing in the shipped path (the C-ABI library, the C++ host classes)
 * links, loads or calls it.
 *
 * Every function restates, in plain C99, the arithmetic of one piece of the
 * reference (wjgra/jpeg-encoder, C++20) and cites the file:line it follows
 * (paths relative to the reference root).  Float work is written so that with
 * `-ffp-contract=off` and no fast-math each `*`, `+` and `/` rounds exactly
 * where the reference's does.
 *
 * Pinning: the reference ships no tests and no golden files.  The whole
 * reference cannot be compiled in this image without a stand-in for <SDL.h>
 * (every TU except bit_stream.cpp includes it transitively), so the pins are:
 *   (1) the whole-image JPEG / decoded-RGB FNV-1a-64 hashes that were recorded
 *       from a build of the reference during the survey of this repository
 *       (tests/golden/reference_hashes.json); this file reproduces all of them;
 *   (2) the cosine-table bits and quantisation tables recorded the same way;
 *   (3) oracle/_ref: the reference's own bit_stream.cpp compiled as-is, which
 *       pins the bit packing / byte stuffing / unstuffing semantics.
 *
 * Build:  gcc -std=c99 -O2 -ffp-contract=off -fPIC -shared jpeg_oracle.c -o libjpeg_oracle.so -lm
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

#define JO_OK 0
#define JO_ERR_CAPACITY 1
#define JO_ERR_BAD_STREAM 2
#define JO_ERR_BAD_HEADER 3
#define JO_ERR_RANGE 4
#define JO_ERR_NO_EOI 5

/* ------------------------------------------------------------------------- */
/* Constants                                                                 */
/* ------------------------------------------------------------------------- */

/* Bits of cosf((2.0f*x+1)*u*pi_f/16.0f), rows u, columns x, as the reference
 * evaluates it inline (jpeg/src/discrete_cosine_transform.cpp:108,120,131,142).
 * jo_cos_table_runtime() recomputes them with this machine's libm so a test can
 * confirm the embedded bits. */
static const uint32_t JO_COS_BITS[8][8] = {
    {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u},
    {0x3f7b14beu, 0x3f54db31u, 0x3f0e39d9u, 0x3e47c5bcu, 0xbe47c5c2u, 0xbf0e39dcu, 0xbf54db32u, 0xbf7b14bfu},
    {0x3f6c835eu, 0x3ec3ef15u, 0xbec3ef18u, 0xbf6c8360u, 0xbf6c835eu, 0xbec3ef0bu, 0x3ec3ef1bu, 0x3f6c835fu},
    {0x3f54db31u, 0xbe47c5c2u, 0xbf7b14bfu, 0xbf0e39d6u, 0x3f0e39ddu, 0x3f7b14beu, 0x3e47c5b1u, 0xbf54db34u},
    {0x3f3504f3u, 0xbf3504f3u, 0xbf3504f1u, 0x3f3504f7u, 0x3f3504f3u, 0xbf3504fbu, 0xbf3504efu, 0x3f3504f4u},
    {0x3f0e39d9u, 0xbf7b14bfu, 0x3e47c5c8u, 0x3f54db2du, 0xbf54db34u, 0xbe47c5bbu, 0x3f7b14bfu, 0xbf0e39e4u},
    {0x3ec3ef15u, 0xbf6c835eu, 0x3f6c835fu, 0xbec3ef25u, 0xbec3ef06u, 0x3f6c835bu, 0xbf6c8362u, 0x3ec3ef25u},
    {0x3e47c5bcu, 0xbf0e39d6u, 0x3f54db2du, 0xbf7b14bdu, 0x3f7b14c1u, 0xbf54db31u, 0x3f0e39e9u, 0xbe47c614u},
};
/* 0.5f * (1/sqrtf(2.0f)), discrete_cosine_transform.cpp:104 */
#define JO_SCALE0_BITS 0x3eb504f3u

static float jo_f32(uint32_t bits) { float f; memcpy(&f, &bits, 4); return f; }
static uint32_t jo_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

static float JO_C[8][8];      /* cos table                        */
static float JO_S[8];         /* scale factors s(u)               */
static float JO_SC[8][8];     /* fl32(s(u) * C[u][x]) for inverse */
static int jo_ready = 0;

static void jo_init(void) {
    if (jo_ready) return;
    for (int u = 0; u < 8; ++u) {
        JO_S[u] = (u == 0) ? jo_f32(JO_SCALE0_BITS) : 0.5f;
        for (int x = 0; x < 8; ++x) JO_C[u][x] = jo_f32(JO_COS_BITS[u][x]);
    }
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) JO_SC[u][x] = JO_S[u] * JO_C[u][x];
    jo_ready = 1;
}

/* Writes the 64 embedded cosine bit patterns (row-major u,x) and the scale-0 bits. */
void jo_cos_table(uint32_t out_bits[64], uint32_t* scale0_bits) {
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) out_bits[8 * u + x] = JO_COS_BITS[u][x];
    *scale0_bits = JO_SCALE0_BITS;
}

/* The same table evaluated with this host's libm, written the way the reference
 * writes it: (2.0f * x + 1) * u * pi_v<float> / 16.0f, left to right in float
 * (discrete_cosine_transform.cpp:108), and 0.5f * (1/sqrt(2.0f)) (:104). */
void jo_cos_table_runtime(uint32_t out_bits[64], uint32_t* scale0_bits) {
    const float pi_f = 3.14159265358979323846f;
    for (int u = 0; u < 8; ++u)
        for (int x = 0; x < 8; ++x) {
            volatile float a = 2.0f * (float)x + 1;
            volatile float b = a * (float)u;
            volatile float c = b * pi_f;
            volatile float d = c / 16.0f;
            out_bits[8 * u + x] = jo_bits(cosf(d));
        }
    volatile float r = 1 / sqrtf(2.0f);
    *scale0_bits = jo_bits(0.5f * r);
}

/* Zigzag scan order: zz[k] = natural[JO_ZIGZAG[k]].  Same walk as
 * entropy_encoder.cpp:15-77 and the literal list at quantiser.cpp:93. */
static const uint8_t JO_ZIGZAG[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void jo_zigzag_order(uint8_t out[64]) { memcpy(out, JO_ZIGZAG, 64); }

/* The zigzag order produced by literally walking the block the way
 * entropy_encoder.cpp:20-75 does (south-west / north-east legs); used by a test
 * to confirm JO_ZIGZAG. */
void jo_zigzag_walk(uint8_t out[64]) {
    size_t x = 0, y = 0, k = 1;
    int sw = 1;
    out[0] = 0;
    while (x + y < 2 * 8 - 3) {
        if (sw) {
            if (x == 7) ++y; else ++x;
            out[k++] = (uint8_t)(x + 8 * y);
            do { --x; ++y; out[k++] = (uint8_t)(x + 8 * y); } while (x != 0 && y != 7);
            sw = 0;
        } else {
            if (y == 7) ++x; else ++y;
            out[k++] = (uint8_t)(x + 8 * y);
            do { ++x; --y; out[k++] = (uint8_t)(x + 8 * y); } while (x != 7 && y != 0);
            sw = 1;
        }
    }
    out[k] = (uint8_t)k; /* entropy_encoder.cpp:75: last entry maps to itself */
}

/* ITU-T T.81 Annex K "typical" Huffman tables in DHT form (counts per code
 * length 1..16, then symbols).  The reference hard-codes the same payload at
 * entropy_encoder.cpp:639-669 and the equivalent code words at :146-613. */
static const uint8_t JO_DC_LUM_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t JO_DC_CHR_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t JO_DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t JO_AC_LUM_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t JO_AC_LUM_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t JO_AC_CHR_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t JO_AC_CHR_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint16_t code[256]; /* indexed by symbol (RRRRSSSS or SSSS) */
    uint8_t len[256];   /* 0 = symbol not in table */
} jo_hufftab;

static jo_hufftab JO_HT[2][2]; /* [0=lum,1=chr][0=dc,1=ac] */
static int jo_ht_ready = 0;

/* Canonical code assignment of T.81 Annex C: yields the code words the
 * reference lists at entropy_encoder.cpp:149-162,166-379,383-396,398-611. */
static void jo_build_hufftab(jo_hufftab* t, const uint8_t bits[16], const uint8_t* vals) {
    memset(t, 0, sizeof(*t));
    unsigned code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l - 1]; ++i, ++k) {
            t->code[vals[k]] = (uint16_t)code++;
            t->len[vals[k]] = (uint8_t)l;
        }
        code <<= 1;
    }
}
/* The tables in use, in the form the header carries them (entropy_encoder.cpp:672-693): for each of the four
 * tables in header order -- lum DC, lum AC, chr DC, chr AC -- the class/id byte, 16 counts, the values: 29, 179,
 * 29 and 179 bytes.  The reference only ever has the Annex K tables (JO_SPEC_STD); the optimised-table mode
 * below (which the reference names as future work, entropy_encoder.cpp:614) switches the spec for one file. */
#define JO_SPEC_BYTES 416
static uint8_t JO_SPEC[JO_SPEC_BYTES];
static int jo_spec_ready = 0;
static int jo_dlut_ready;
static void jo_spec_std(uint8_t* o) {
    size_t p = 0;
    o[p++] = 0x00; memcpy(o + p, JO_DC_LUM_BITS, 16); p += 16; memcpy(o + p, JO_DC_VALS, 12); p += 12;
    o[p++] = 0x10; memcpy(o + p, JO_AC_LUM_BITS, 16); p += 16; memcpy(o + p, JO_AC_LUM_VALS, 162); p += 162;
    o[p++] = 0x01; memcpy(o + p, JO_DC_CHR_BITS, 16); p += 16; memcpy(o + p, JO_DC_VALS, 12); p += 12;
    o[p++] = 0x11; memcpy(o + p, JO_AC_CHR_BITS, 16); p += 16; memcpy(o + p, JO_AC_CHR_VALS, 162); p += 162;
}
static const size_t JO_SPEC_AT[4] = {0, 29, 208, 237};   /* lum DC, lum AC, chr DC, chr AC */
static void jo_init_huff(void) {
    if (jo_ht_ready) return;
    if (!jo_spec_ready) { jo_spec_std(JO_SPEC); jo_spec_ready = 1; }
    jo_build_hufftab(&JO_HT[0][0], JO_SPEC + JO_SPEC_AT[0] + 1, JO_SPEC + JO_SPEC_AT[0] + 17);
    jo_build_hufftab(&JO_HT[0][1], JO_SPEC + JO_SPEC_AT[1] + 1, JO_SPEC + JO_SPEC_AT[1] + 17);
    jo_build_hufftab(&JO_HT[1][0], JO_SPEC + JO_SPEC_AT[2] + 1, JO_SPEC + JO_SPEC_AT[2] + 17);
    jo_build_hufftab(&JO_HT[1][1], JO_SPEC + JO_SPEC_AT[3] + 1, JO_SPEC + JO_SPEC_AT[3] + 17);
    jo_ht_ready = 1;
}
/* spec = NULL: back to the Annex K tables.  Not thread safe (like jo_set_mapper). */
void jo_use_huffman_spec(const uint8_t* spec) {
    if (spec) memcpy(JO_SPEC, spec, JO_SPEC_BYTES); else jo_spec_std(JO_SPEC);
    jo_spec_ready = 1;
    jo_ht_ready = 0;
    jo_dlut_ready = 0;
}

/* Exports the four code tables: code[4][256] / len[4][256] in the order
 * lum-DC, lum-AC, chr-DC, chr-AC. */
void jo_huffman_tables(uint16_t* code, uint8_t* len) {
    jo_init_huff();
    for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 2; ++a) {
            memcpy(code + (2 * c + a) * 256, JO_HT[c][a].code, 512);
            memcpy(len + (2 * c + a) * 256, JO_HT[c][a].len, 256);
        }
}

/* ------------------------------------------------------------------------- */
/* Quantiser tables -- quantiser.cpp:3-55                                     */
/* ------------------------------------------------------------------------- */
static const uint16_t JO_BASE_LUM[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint16_t JO_BASE_CHR[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

void jo_quant_tables(int quality, uint16_t lum[64], uint16_t chr[64]) {
    if (quality < 1) quality = 1;              /* quantiser.cpp:6-11 */
    else if (quality > 100) quality = 100;
    /* :25 -- integer arithmetic, then converted to float */
    float S = (float)((quality < 50) ? 5000 / quality : 200 - 2 * quality);
    for (int i = 0; i < 64; ++i) {
        /* :27-29 float: (S*base + 50)/100, floor, narrowed to uint16 */
        float l = floorf((S * (float)JO_BASE_LUM[i] + 50.0f) / 100.0f);
        float c = floorf((S * (float)JO_BASE_CHR[i] + 50.0f) / 100.0f);
        lum[i] = (uint16_t)l;
        chr[i] = (uint16_t)c;
        if (lum[i] == 0) lum[i] = 1;           /* :30-32 */
        if (chr[i] == 0) chr[i] = 1;           /* :51-53 */
    }
}

/* ------------------------------------------------------------------------- */
/* Per-block components                                                       */
/* ------------------------------------------------------------------------- */

/* colour_mapping.cpp:48-51 */
static uint8_t jo_map_to_range(double in) {
    double r = floor(in + 0.5);
    if (r < 0.0) r = 0.0;
    if (r > 255.0) r = 255.0;
    return (uint8_t)r;
}

/* colour_mapping.cpp:52-60; double, left-to-right */
void jo_rgb_to_ycbcr_px(uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]) {
    out[0] = jo_map_to_range(0 + 0.299 * r + 0.587 * g + 0.114 * b);
    out[1] = jo_map_to_range(128 - 0.168736 * r - 0.331264 * g + 0.5 * b);
    out[2] = jo_map_to_range(128 + 0.5 * r - 0.418688 * g - 0.081312 * b);
}

/* colour_mapping.cpp:77-85 */
void jo_ycbcr_to_rgb_px(uint8_t Y, uint8_t Cb, uint8_t Cr, uint8_t out[3]) {
    out[0] = jo_map_to_range(Y + 1.402 * (Cr - 128));
    out[1] = jo_map_to_range(Y - 0.344136 * (Cb - 128) - 0.714136 * (Cr - 128));
    out[2] = jo_map_to_range(Y + 1.772 * (Cb - 128));
}

/* Bulk versions for exhaustive sweeps: n pixels, interleaved triples. */
/* Colour mapper used by the whole-image functions below: 0 = RGBToYCbCrMapper
 * (colour_mapping.cpp:44-97, what BaselineEncoder composes), 1 = RGBToRGBMapper
 * (colour_mapping.cpp:16-42: components pass through, all three are "luminance").
 * The reference hashes were recorded with mapper 0 only: mapper 1 is restated from
 * the source and not pinned by any reference output. */
static int jo_mapper = 0;
void jo_set_mapper(int mapper) { jo_mapper = mapper ? 1 : 0; }
static int jo_is_lum(int ch) { return jo_mapper ? 1 : ch == 0; }

void jo_rgb_to_ycbcr(const uint8_t* rgb, size_t n, uint8_t* ycc) {
    for (size_t i = 0; i < n; ++i) jo_rgb_to_ycbcr_px(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], ycc + 3 * i);
}
void jo_ycbcr_to_rgb(const uint8_t* ycc, size_t n, uint8_t* rgb) {
    for (size_t i = 0; i < n; ++i) jo_ycbcr_to_rgb_px(ycc[3 * i], ycc[3 * i + 1], ycc[3 * i + 2], rgb + 3 * i);
}

/* Forward DCT of one level-shifted channel: discrete_cosine_transform.cpp:77-88
 * (driver), :103-112 (row pass; result stored transposed at 8u+y), :114-124
 * (column pass; result at 8v+u).  Every product and sum rounds to float. */
void jo_fdct(const int8_t src[64], float out[64]) {
    jo_init();
    float row[64];
    for (int u = 0; u < 8; ++u)
        for (int y = 0; y < 8; ++y) {
            float acc = 0;
            for (int x = 0; x < 8; ++x) acc += JO_C[u][x] * (float)src[x + 8 * y];
            row[8 * u + y] = JO_S[u] * acc;
        }
    for (int v = 0; v < 8; ++v)
        for (int u = 0; u < 8; ++u) {
            float acc = 0;
            for (int y = 0; y < 8; ++y) acc += JO_C[v][y] * row[y + 8 * u];
            out[8 * v + u] = JO_S[v] * acc;
        }
}

/* Inverse DCT: discrete_cosine_transform.cpp:90-101 (driver), :126-135 (row),
 * :136-156 (column, clamp, truncating conversion). */
void jo_idct(const float src[64], int8_t out[64]) {
    jo_init();
    float tmp[64];
    for (int x = 0; x < 8; ++x)
        for (int v = 0; v < 8; ++v) {
            float acc = 0;
            for (int u = 0; u < 8; ++u) acc += JO_SC[u][x] * src[u + 8 * v];
            tmp[8 * x + v] = acc;
        }
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            float acc = 0;
            for (int v = 0; v < 8; ++v) acc += JO_SC[v][y] * tmp[v + 8 * x];
            int8_t o;
            if (acc < -128.0f) o = -128;
            else if (acc > 127.0f) o = 127;
            else o = (int8_t)acc;
            out[8 * y + x] = o;
        }
}

/* quantiser.cpp:57-70: float divide, then `0.5 +` and floor in double. */
void jo_quantise(const float dct[64], const uint16_t q[64], int16_t out[64]) {
    for (int i = 0; i < 64; ++i) {
        float quot = dct[i] / (float)q[i];
        out[i] = (int16_t)floor(0.5 + (double)quot);
    }
}

/* quantiser.cpp:72-85 */
void jo_dequantise(const int16_t in[64], const uint16_t q[64], float out[64]) {
    for (int i = 0; i < 64; ++i) out[i] = (float)in[i] * (float)q[i];
}

/* Fetch MCU (bx,by) with right/bottom edge replication: block_grid.cpp:8-28. */
static void jo_fetch_block(const uint8_t* rgb, int w, int h, int bx, int by, uint8_t blk[64][3]) {
    for (int row = 0; row < 8; ++row) {
        int yy = by * 8 + row; if (yy > h - 1) yy = h - 1;
        for (int col = 0; col < 8; ++col) {
            int xx = bx * 8 + col; if (xx > w - 1) xx = w - 1;
            const uint8_t* p = rgb + 3 * ((size_t)yy * (size_t)w + (size_t)xx);
            blk[8 * row + col][0] = p[0]; blk[8 * row + col][1] = p[1]; blk[8 * row + col][2] = p[2];
        }
    }
}

/* RGB image -> quantised, zig-zagged coefficient blocks laid out
 * [mcu][channel][64] in raster MCU order: block_grid.cpp:8-58,
 * colour_mapping.cpp:44-67, discrete_cosine_transform.cpp:3-6,13-19,77-124,
 * quantiser.cpp:57-70, entropy_encoder.cpp:15-77; loop of encoder.cpp:21-27. */
int jo_image_to_coefs(const uint8_t* rgb, int w, int h, const uint16_t lumq[64], const uint16_t chrq[64],
                      int16_t* coefs) {
    if (w <= 0 || h <= 0) return JO_ERR_RANGE;
    int bw = (w + 7) / 8, bh = (h + 7) / 8;
    uint8_t blk[64][3];
    for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
            jo_fetch_block(rgb, w, h, bx, by, blk);
            uint8_t ycc[3][64];
            for (int i = 0; i < 64; ++i) {
                uint8_t o[3];
                if (jo_mapper) { o[0] = blk[i][0]; o[1] = blk[i][1]; o[2] = blk[i][2]; }
                else jo_rgb_to_ycbcr_px(blk[i][0], blk[i][1], blk[i][2], o);
                ycc[0][i] = o[0]; ycc[1][i] = o[1]; ycc[2][i] = o[2];
            }
            for (int ch = 0; ch < 3; ++ch) {
                int8_t shifted[64];
                float dct[64];
                int16_t q[64];
                for (int i = 0; i < 64; ++i) shifted[i] = (int8_t)(ycc[ch][i] - 128);
                jo_fdct(shifted, dct);
                jo_quantise(dct, jo_is_lum(ch) ? lumq : chrq, q);
                int16_t* dst = coefs + ((size_t)(by * bw + bx) * 3 + ch) * 64;
                for (int k = 0; k < 64; ++k) dst[k] = q[JO_ZIGZAG[k]];
            }
        }
    return JO_OK;
}

/* ------------------------------------------------------------------------- */
/* Bit writer -- bit_stream.cpp:44-81 (MSB first), :101-105 (zero padding)    */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint8_t* buf;
    size_t cap, len;     /* committed bytes */
    uint32_t acc;        /* pending bits, right aligned */
    int nacc;            /* number of pending bits (0..7 after flush) */
    int overflow;
} jo_bw;

static void jo_bw_put(jo_bw* w, unsigned value, int nbits) {
    if (nbits == 0) return;
    w->acc = (w->acc << nbits) | (value & ((1u << nbits) - 1u));
    w->nacc += nbits;
    while (w->nacc >= 8) {
        uint8_t byte = (uint8_t)(w->acc >> (w->nacc - 8));
        if (w->len < w->cap) w->buf[w->len] = byte; else w->overflow = 1;
        w->len++;
        w->nacc -= 8;
    }
    w->acc &= (1u << w->nacc) - 1u;
}

static int jo_bit_width(unsigned v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

/* Number of bits one block-channel contributes, and (optionally) the bits
 * themselves.  entropy_encoder.cpp:94-120 (run-length pass incl. the trailing
 * ZRL removal), :715-730 (DC), :732-765 (AC).  zz is in zigzag order. */
static uint32_t* jo_hist = NULL;   /* [2 classes][2: DC, AC][256] symbol counts, filled while non-NULL */
static int jo_encode_block(const int16_t zz[64], int16_t* last_dc, int is_lum, jo_bw* w, uint32_t* nbits_out) {
    jo_init_huff();
    const jo_hufftab* dc = &JO_HT[is_lum ? 0 : 1][0];
    const jo_hufftab* ac = &JO_HT[is_lum ? 0 : 1][1];
    uint32_t nbits = 0;
    int diff = (int16_t)(zz[0] - *last_dc);
    *last_dc = zz[0];
    unsigned amp = (unsigned)(diff > 0 ? diff : -diff);
    int ssss = jo_bit_width(amp);
    if (ssss > 11) return JO_ERR_RANGE;
    if (jo_hist) jo_hist[((is_lum ? 0 : 1) * 2 + 0) * 256 + ssss]++;
    if (w) jo_bw_put(w, dc->code[ssss], dc->len[ssss]);
    nbits += dc->len[ssss];
    if (ssss > 0) {
        if (w) jo_bw_put(w, diff > 0 ? amp : ~amp, ssss);
        nbits += (uint32_t)ssss;
    }
    /* run-length list */
    int runs[64], vals[64], n = 0, zc = 0;
    for (int k = 1; k < 64; ++k) {
        if (zc == 15 || zz[k] != 0) { runs[n] = zc; vals[n] = zz[k]; ++n; zc = 0; }
        else ++zc;
    }
    if (zz[63] == 0) {
        while (n > 0 && vals[n - 1] == 0) --n;
        runs[n] = 0; vals[n] = 0; ++n;
    }
    for (int i = 0; i < n; ++i) {
        int v = vals[i];
        unsigned a = (unsigned)(v > 0 ? v : -v);
        int s = jo_bit_width(a);
        unsigned sym;
        if (s == 0) {
            if (runs[i] == 0) sym = 0x00;        /* EOB */
            else if (runs[i] == 15) sym = 0xF0;  /* ZRL */
            else return JO_ERR_RANGE;            /* :748 */
        } else {
            if (s > 10) return JO_ERR_RANGE;
            sym = (unsigned)(runs[i] << 4) | (unsigned)s;
        }
        if (jo_hist) jo_hist[((is_lum ? 0 : 1) * 2 + 1) * 256 + sym]++;
        if (w) jo_bw_put(w, ac->code[sym], ac->len[sym]);
        nbits += ac->len[sym];
        if (s > 0) {
            if (w) jo_bw_put(w, v > 0 ? a : ~a, s);
            nbits += (uint32_t)s;
        }
    }
    if (nbits_out) *nbits_out = nbits;
    return JO_OK;
}

/* Bit count of every block-channel ([mcu][ch] order) with the DC prediction
 * chain of encoder.cpp:20 (per channel, starts at 0, never reset). */
int jo_coefs_to_bit_lengths(const int16_t* coefs, size_t n_mcu, uint32_t* lengths) {
    int16_t last[3] = {0, 0, 0};
    for (size_t m = 0; m < n_mcu; ++m)
        for (int ch = 0; ch < 3; ++ch) {
            int rc = jo_encode_block(coefs + (m * 3 + (size_t)ch) * 64, &last[ch], jo_is_lum(ch), NULL,
                                     &lengths[m * 3 + (size_t)ch]);
            if (rc) return rc;
        }
    return JO_OK;
}

/* Entropy-coded scan segment of an image given its coefficient blocks:
 * Huffman bits of all blocks (encoder.cpp:21-27), byte stuffing of every
 * committed 0xFF (bit_stream.cpp:127-134, intended semantics: a 0x00 after
 * each), then zero-bit padding of the last partial byte (encoder.cpp:31).
 * The pending partial byte is not visible to the stuffing pass, and cannot be
 * 0xFF after zero padding.  No EOI marker is written here. */
int jo_coefs_to_scan(const int16_t* coefs, size_t n_mcu, uint8_t* out, size_t cap, size_t* out_len,
                     uint64_t* total_bits) {
    jo_bw w = {0};
    /* unstuffed stream first, then stuff: mirrors the reference's two steps */
    size_t raw_cap = n_mcu * 3 * 208 + 16;
    uint8_t* raw = (uint8_t*)malloc(raw_cap);
    if (!raw) return JO_ERR_CAPACITY;
    w.buf = raw; w.cap = raw_cap;
    int16_t last[3] = {0, 0, 0};
    uint64_t bits = 0;
    for (size_t m = 0; m < n_mcu; ++m)
        for (int ch = 0; ch < 3; ++ch) {
            uint32_t nb = 0;
            int rc = jo_encode_block(coefs + (m * 3 + (size_t)ch) * 64, &last[ch], jo_is_lum(ch), &w, &nb);
            if (rc) { free(raw); return rc; }
            bits += nb;
        }
    if (total_bits) *total_bits = bits;
    size_t o = 0;
    int over = 0;
    for (size_t i = 0; i < w.len; ++i) {
        if (o < cap) out[o] = raw[i]; else over = 1;
        ++o;
        if (raw[i] == 0xFF) { if (o < cap) out[o] = 0x00; else over = 1; ++o; }
    }
    if (w.nacc > 0) {
        uint8_t byte = (uint8_t)(w.acc << (8 - w.nacc));
        if (o < cap) out[o] = byte; else over = 1;
        ++o;
    }
    free(raw);
    *out_len = o;
    return over ? JO_ERR_CAPACITY : JO_OK;
}

/* ------------------------------------------------------------------------- */
/* Header -- encoder.cpp:74-138, quantiser.cpp:87-102,                        */
/*           entropy_encoder.cpp:636-694.  Always 607 bytes.                  */
/* ------------------------------------------------------------------------- */
#define JO_HEADER_BYTES 607

static size_t jo_put8(uint8_t* o, size_t p, unsigned v) { o[p] = (uint8_t)v; return p + 1; }
static size_t jo_put16(uint8_t* o, size_t p, unsigned v) { o[p] = (uint8_t)(v >> 8); o[p + 1] = (uint8_t)v; return p + 2; }

size_t jo_write_header(uint8_t out[JO_HEADER_BYTES], int w, int h, const uint16_t lumq[64], const uint16_t chrq[64]) {
    size_t p = 0;
    p = jo_put16(out, p, 0xFFD8);                       /* SOI  */
    p = jo_put16(out, p, 0xFFE0); p = jo_put16(out, p, 16); /* APP0 */
    p = jo_put8(out, p, 'J'); p = jo_put8(out, p, 'F'); p = jo_put8(out, p, 'I'); p = jo_put8(out, p, 'F');
    p = jo_put8(out, p, 0);
    p = jo_put16(out, p, 0x0102); p = jo_put8(out, p, 0);
    p = jo_put16(out, p, 0x0010); p = jo_put16(out, p, 0x0010); p = jo_put16(out, p, 0x0000);
    p = jo_put16(out, p, 0xFFDB); p = jo_put16(out, p, 2 + 2 * 65);   /* DQT */
    p = jo_put8(out, p, 0x00);
    for (int k = 0; k < 64; ++k) p = jo_put8(out, p, lumq[JO_ZIGZAG[k]]);  /* truncated to 8 bits */
    p = jo_put8(out, p, 0x01);
    for (int k = 0; k < 64; ++k) p = jo_put8(out, p, chrq[JO_ZIGZAG[k]]);
    p = jo_put16(out, p, 0xFFC0); p = jo_put16(out, p, 17);            /* SOF0 */
    p = jo_put8(out, p, 8);
    p = jo_put16(out, p, (unsigned)h); p = jo_put16(out, p, (unsigned)w);
    p = jo_put8(out, p, 3);
    p = jo_put8(out, p, 1); p = jo_put8(out, p, 0x11); p = jo_put8(out, p, 0);
    p = jo_put8(out, p, 2); p = jo_put8(out, p, 0x11); p = jo_put8(out, p, 1);
    p = jo_put8(out, p, 3); p = jo_put8(out, p, 0x11); p = jo_put8(out, p, 1);
    p = jo_put16(out, p, 0xFFC4);                                       /* DHT */
    p = jo_put16(out, p, 2 + 4 * 17 + 12 + 12 + 162 + 162);
    jo_init_huff();
    for (int i = 0; i < JO_SPEC_BYTES; ++i) p = jo_put8(out, p, JO_SPEC[i]);
    p = jo_put16(out, p, 0xFFDA); p = jo_put16(out, p, 12);            /* SOS */
    p = jo_put8(out, p, 3);
    p = jo_put8(out, p, 1); p = jo_put8(out, p, 0x00);
    p = jo_put8(out, p, 2); p = jo_put8(out, p, 0x11);
    p = jo_put8(out, p, 3); p = jo_put8(out, p, 0x11);
    p = jo_put16(out, p, 0x003F); p = jo_put8(out, p, 0x00);
    return p;
}

/* Upper bound of the JPEG size for a w x h image: header + worst-case scan
 * (1660 bits per block-channel, every byte stuffed) + pad byte + EOI. */
size_t jo_max_jpeg_bytes(int w, int h) {
    size_t n_mcu = (size_t)((w + 7) / 8) * (size_t)((h + 7) / 8);
    return JO_HEADER_BYTES + 2 * ((n_mcu * 3 * 1660 + 7) / 8) + 3;
}

/* Whole encode: encoder.cpp:14-42. */
int jo_encode(const uint8_t* rgb, int w, int h, int quality, uint8_t* out, size_t cap, size_t* out_len) {
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return JO_ERR_RANGE;
    if (cap < JO_HEADER_BYTES + 2) return JO_ERR_CAPACITY;
    uint16_t lumq[64], chrq[64];
    jo_quant_tables(quality, lumq, chrq);
    size_t n_mcu = (size_t)((w + 7) / 8) * (size_t)((h + 7) / 8);
    int16_t* coefs = (int16_t*)malloc(n_mcu * 3 * 64 * sizeof(int16_t));
    if (!coefs) return JO_ERR_CAPACITY;
    int rc = jo_image_to_coefs(rgb, w, h, lumq, chrq, coefs);
    if (rc) { free(coefs); return rc; }
    size_t p = jo_write_header(out, w, h, lumq, chrq);
    size_t scan_len = 0;
    rc = jo_coefs_to_scan(coefs, n_mcu, out + p, cap - p - 2, &scan_len, NULL);
    free(coefs);
    if (rc) return rc;
    p += scan_len;
    p = jo_put16(out, p, 0xFFD9);
    *out_len = p;
    return JO_OK;
}

/* ------------------------------------------------------------------------- */
/* Per-image Huffman tables.  The reference has none ("Issue: implement        */
/* frequency analysis to generate tailored Huffman tables - requires two       */
/* passes in encoder", entropy_encoder.cpp:614); this is the specification of  */
/* this build's optional mode, which the GPU code must reproduce bit for bit.  */
/* Parity unpinned: nothing of the reference covers it; the files are checked  */
/* against an independent JPEG decoder by the tests.                           */
/*                                                                             */
/* For each of the four tables, over its symbol set (DC: 0..11; AC: the 162    */
/* symbols of the Annex K lists) in ascending symbol value, id = 0..n-1:       */
/*  1. weight = occurrences + 1 (every symbol keeps a code, so the header      */
/*     keeps its 607 bytes and its layout); one more leaf, id n, weight 0: it  */
/*     ends up with the longest code and keeps the all-ones code word unused   */
/*     (T.81 K.2 does the same with symbol 256).                               */
/*  2. leaves sorted by (weight, id descending); Huffman tree by the two-queue */
/*     method: of the heads of the leaf queue and of the queue of merged       */
/*     nodes (in creation order) the lighter one is taken, a leaf on a tie.    */
/*  3. code length = depth; lengths above 16 (AC) or 11 (DC, like the Annex K   */
/*     codes) are brought down as in T.81 Figure K.3; the extra leaf's count   */
/*     leaves the longest length in use.                                       */
/*  4. symbols ordered by (depth, id) take the lengths in ascending order:     */
/*     that order is the table's value list (T.81 Annex C gives the codes).    */
/* ------------------------------------------------------------------------- */
static int jo_sym_set(int is_ac, uint8_t* syms) {
    int n = 0;
    if (!is_ac) { for (int s = 0; s < 12; ++s) syms[n++] = (uint8_t)s; return n; }
    for (int s = 0; s < 256; ++s) {
        int ss = s & 15;
        if ((ss >= 1 && ss <= 10) || s == 0x00 || s == 0xF0) syms[n++] = (uint8_t)s;
    }
    return n;   /* 162 */
}

static void jo_optimal_table(const uint32_t* freq /*[256]*/, int is_ac, uint8_t bits_out[16], uint8_t* vals_out) {
    uint8_t syms[163];
    const int n = jo_sym_set(is_ac, syms), N = n + 1;
    uint64_t key[163];            /* weight << 9 | (511 - id): ascending = (weight, id descending) */
    for (int i = 0; i < n; ++i) key[i] = ((uint64_t)freq[syms[i]] + 1) << 9 | (uint64_t)(511 - i);
    key[n] = (uint64_t)(511 - n);
    int order[163];               /* leaves by ascending key */
    for (int i = 0; i < N; ++i) {
        int rank = 0;
        for (int j = 0; j < N; ++j) rank += key[j] < key[i];
        order[rank] = i;
    }
    /* nodes 0..N-1: leaves in sorted order; N..2N-2: merged nodes in creation order */
    uint64_t weight[2 * 163];
    int parent[2 * 163];
    for (int i = 0; i < N; ++i) weight[i] = key[order[i]] >> 9;
    int lh = 0, ih = N, made = N;
    for (int m = 0; m < N - 1; ++m) {
        int pick[2];
        for (int k = 0; k < 2; ++k) {
            if (lh < N && (ih >= made || weight[lh] <= weight[ih])) pick[k] = lh++;
            else pick[k] = ih++;
        }
        weight[made] = weight[pick[0]] + weight[pick[1]];
        parent[pick[0]] = parent[pick[1]] = made;
        ++made;
    }
    int depth[2 * 163];
    depth[made - 1] = 0;
    for (int v = made - 2; v >= 0; --v) depth[v] = depth[parent[v]] + 1;
    int count[2 * 163 + 1];
    memset(count, 0, sizeof(count));
    int leaf_depth[163];          /* by id */
    for (int i = 0; i < N; ++i) { leaf_depth[order[i]] = depth[i]; count[depth[i]]++; }
    const int limit = is_ac ? 16 : 11;        /* DC codes stay within the 11 bits of the Annex K ones */
    for (int i = 2 * 163; i > limit; --i) {   /* T.81 Figure K.3 */
        while (count[i] > 0) {
            int j = i - 2;
            while (count[j] == 0) --j;
            count[i] -= 2; count[i - 1] += 1; count[j + 1] += 2; count[j] -= 1;
        }
    }
    { int i = limit; while (count[i] == 0) --i; count[i]--; }
    for (int l = 1; l <= 16; ++l) bits_out[l - 1] = (uint8_t)count[l];
    uint32_t skey[162];
    for (int i = 0; i < n; ++i) skey[i] = (uint32_t)leaf_depth[i] << 8 | (uint32_t)i;
    for (int i = 0; i < n; ++i) {
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += skey[j] < skey[i];
        vals_out[rank] = syms[i];
    }
}

/* The optimal tables of an image's coefficient blocks as a table spec (JO_SPEC_BYTES bytes). */
int jo_optimal_huffman_spec(const int16_t* coefs, size_t n_mcu, uint8_t* spec) {
    uint32_t* hist = (uint32_t*)calloc(4 * 256, sizeof(uint32_t));
    if (!hist) return JO_ERR_CAPACITY;
    int16_t last[3] = {0, 0, 0};
    jo_hist = hist;
    int rc = JO_OK;
    for (size_t m = 0; m < n_mcu && !rc; ++m)
        for (int ch = 0; ch < 3 && !rc; ++ch)
            rc = jo_encode_block(coefs + (m * 3 + (size_t)ch) * 64, &last[ch], jo_is_lum(ch), NULL, NULL);
    jo_hist = NULL;
    if (!rc) {
        static const uint8_t ids[4] = {0x00, 0x10, 0x01, 0x11};
        for (int t = 0; t < 4; ++t) {
            uint8_t* o = spec + JO_SPEC_AT[t];
            o[0] = ids[t];
            /* header order lum DC, lum AC, chr DC, chr AC; hist order [class][dc/ac] */
            jo_optimal_table(hist + ((t >> 1) * 2 + (t & 1)) * 256, t & 1, o + 1, o + 17);
        }
    }
    free(hist);
    return rc;
}

/* jo_encode with the image's own optimal tables in the header and in the scan. */
int jo_encode_optimised(const uint8_t* rgb, int w, int h, int quality, uint8_t* out, size_t cap, size_t* out_len) {
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return JO_ERR_RANGE;
    if (cap < JO_HEADER_BYTES + 2) return JO_ERR_CAPACITY;
    uint16_t lumq[64], chrq[64];
    jo_quant_tables(quality, lumq, chrq);
    size_t n_mcu = (size_t)((w + 7) / 8) * (size_t)((h + 7) / 8);
    int16_t* coefs = (int16_t*)malloc(n_mcu * 3 * 64 * sizeof(int16_t));
    if (!coefs) return JO_ERR_CAPACITY;
    int rc = jo_image_to_coefs(rgb, w, h, lumq, chrq, coefs);
    uint8_t spec[JO_SPEC_BYTES];
    if (!rc) rc = jo_optimal_huffman_spec(coefs, n_mcu, spec);
    if (rc) { free(coefs); return rc; }
    jo_use_huffman_spec(spec);
    size_t p = jo_write_header(out, w, h, lumq, chrq);
    size_t scan_len = 0;
    rc = jo_coefs_to_scan(coefs, n_mcu, out + p, cap - p - 2, &scan_len, NULL);
    jo_use_huffman_spec(NULL);
    free(coefs);
    if (rc) return rc;
    p += scan_len;
    p = jo_put16(out, p, 0xFFD9);
    *out_len = p;
    return JO_OK;
}

/* ------------------------------------------------------------------------- */
/* Decode                                                                     */
/* ------------------------------------------------------------------------- */

/* Strict header check of encoder.cpp:140-283: every constant field must match,
 * DQT (134 B) and DHT (420 B) are skipped by fixed offsets (:188, :240).
 * Returns 0 or the 1-based index of the first failing check (the order of the
 * reference's throw statements); hdr_w and hdr_h receive the SOF0 dimensions. */
int jo_check_header(const uint8_t* d, size_t len, int* hdr_w, int* hdr_h) {
    if (len < JO_HEADER_BYTES) return -1;
    size_t p = 0;
    int chk = 0;
#define RD8() (d[p++])
#define RD16() (p += 2, (unsigned)((d[p - 2] << 8) | d[p - 1]))
#define EXPECT(cond) do { ++chk; if (!(cond)) return chk; } while (0)
    EXPECT(RD16() == 0xFFD8);
    EXPECT(RD16() == 0xFFE0);
    { size_t s = p; unsigned l = RD16();
      EXPECT(RD8() == 'J'); EXPECT(RD8() == 'F'); EXPECT(RD8() == 'I'); EXPECT(RD8() == 'F'); EXPECT(RD8() == 0);
      EXPECT(RD16() == 0x0102); EXPECT(RD8() == 0); EXPECT(RD16() == 0x0010); EXPECT(RD16() == 0x0010);
      EXPECT(RD16() == 0x0000); EXPECT(l == p - s); }
    p += 2 + 2 + 2 * 65;
    EXPECT(RD16() == 0xFFC0);
    { size_t s = p; unsigned l = RD16();
      EXPECT(RD8() == 8);
      *hdr_h = (int)RD16(); *hdr_w = (int)RD16();
      EXPECT(RD8() == 3);
      EXPECT(RD8() == 1); EXPECT(RD8() == 0x11); EXPECT(RD8() == 0);
      EXPECT(RD8() == 2); EXPECT(RD8() == 0x11); EXPECT(RD8() == 1);
      EXPECT(RD8() == 3); EXPECT(RD8() == 0x11); EXPECT(RD8() == 1);
      EXPECT(l == p - s); }
    p += 2 + 2 + 4 + 412;
    EXPECT(RD16() == 0xFFDA);
    { size_t s = p; unsigned l = RD16();
      EXPECT(RD8() == 3);
      EXPECT(RD8() == 1); EXPECT(RD8() == 0x00);
      EXPECT(RD8() == 2); EXPECT(RD8() == 0x11);
      EXPECT(RD8() == 3); EXPECT(RD8() == 0x11);
      EXPECT(RD16() == 0x003F); EXPECT(RD8() == 0);
      EXPECT(l == p - s); }
#undef RD8
#undef RD16
#undef EXPECT
    return (p == JO_HEADER_BYTES) ? 0 : -2;
}

/* bit_stream.cpp:136-147: from the first scan byte up to (not including) the
 * last two bytes, drop the byte after each 0xFF when it is 0x00; when it is not
 * 0x00 both bytes are kept and the scan resumes after them.  In place; returns
 * the new length. */
size_t jo_remove_stuffing(uint8_t* d, size_t from, size_t len) {
    size_t r = from, wpos = from;
    if (len < from + 2) return len;
    size_t end = len - 2;
    while (r < end) {
        uint8_t b = d[r++];
        d[wpos++] = b;
        if (b == 0xFF) {
            /* the reference dereferences the next byte even when it is the first
             * EOI byte; that byte is then never erased because it is 0xFF */
            if (d[r] == 0x00) { ++r; }
            else { if (r < end) d[wpos++] = d[r]; else break; ++r; }
        }
    }
    /* copy whatever is left (normally the two EOI bytes) */
    while (r < len) d[wpos++] = d[r++];
    return wpos;
}

typedef struct { const uint8_t* d; size_t len; size_t byte; int bit; int eof; } jo_br;

/* bit_stream.cpp:107-112 */
static int jo_br_bit(jo_br* r) {
    if (r->byte >= r->len) { r->eof = 1; return 0; }
    int b = (r->d[r->byte] >> (7 - r->bit)) & 1;
    if (++r->bit == 8) { r->bit = 0; r->byte++; }
    return b;
}

/* Next 16 bits without consuming them (zeros past the end). */
static unsigned jo_br_peek16(const jo_br* r) {
    uint32_t v = 0;
    for (int i = 0; i < 3; ++i) {
        size_t p = r->byte + (size_t)i;
        v = (v << 8) | (p < r->len ? r->d[p] : 0u);
    }
    return (v >> (8 - r->bit)) & 0xFFFFu;
}
static void jo_br_skip(jo_br* r, int n) {
    size_t bits = (size_t)r->bit + (size_t)n;
    r->byte += bits >> 3;
    r->bit = (int)(bits & 7);
    if (r->byte > r->len || (r->byte == r->len && r->bit > 0)) r->eof = 1;
}

/* 16-bit prefix -> (length << 8) | symbol, 0 when no code word matches. */
static uint16_t JO_DLUT[2][2][65536];
static void jo_init_dlut(void) {
    if (jo_dlut_ready) return;
    jo_init_huff();
    for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 2; ++a) {
            memset(JO_DLUT[c][a], 0, sizeof(JO_DLUT[c][a]));
            for (int s = 0; s < 256; ++s) {
                int l = JO_HT[c][a].len[s];
                if (!l) continue;
                unsigned base = (unsigned)JO_HT[c][a].code[s] << (16 - l);
                for (unsigned k = 0; k < (1u << (16 - l)); ++k)
                    JO_DLUT[c][a][base + k] = (uint16_t)((l << 8) | s);
            }
        }
    jo_dlut_ready = 1;
}

/* The reference grows a candidate code one bit at a time from two bits until it
 * equals a code word of that length (entropy_encoder.cpp:767-778, :810-828);
 * for a prefix code whose shortest word has two bits that is the unique code
 * word that prefixes the stream, which the table lookup returns. */
static int jo_decode_symbol(jo_br* r, const uint16_t* lut) {
    unsigned e = lut[jo_br_peek16(r)];
    if (!e) return -1;
    jo_br_skip(r, (int)(e >> 8));
    if (r->eof) return -1;
    return (int)(e & 0xFF);
}

/* entropy_encoder.cpp:786-806 and :845-864 */
static int jo_read_amplitude(jo_br* r, int ssss) {
    unsigned mask = (1u << ssss) - 1u;
    if (jo_br_bit(r)) {
        unsigned a = 1;
        for (int i = 1; i < ssss; ++i) a = (a << 1) | (unsigned)jo_br_bit(r);
        return (int)(mask & a);
    } else {
        unsigned c = 0;
        for (int i = 1; i < ssss; ++i) c = (c << 1) | (unsigned)jo_br_bit(r);
        return -(int)(mask & ~c);
    }
}

/* One block-channel: entropy_encoder.cpp:703-712 (symbol loop), :122-144
 * (run-length removal -- it iterates over all entries but the LAST one, so when
 * a block ends without EOB its final coefficient is dropped), lastDC chain. */
static int jo_decode_block(jo_br* r, int is_lum, int16_t* last_dc, int16_t zz[64]) {
    const uint16_t* dc = JO_DLUT[is_lum ? 0 : 1][0];
    const uint16_t* ac = JO_DLUT[is_lum ? 0 : 1][1];
    int s = jo_decode_symbol(r, dc);
    if (s < 0) return JO_ERR_BAD_STREAM;
    int diff = (s == 0) ? 0 : jo_read_amplitude(r, s);
    int runs[64], vals[64], n = 0, processed = 0;
    do {
        int sym = jo_decode_symbol(r, ac);
        if (sym < 0) return JO_ERR_BAD_STREAM;
        int run, val;
        if (sym == 0x00) { run = 0; val = 0; }
        else if (sym == 0xF0) { run = 15; val = 0; }
        else {
            run = sym >> 4;
            int ss = sym & 15;
            if (ss == 0) return JO_ERR_BAD_STREAM;
            val = jo_read_amplitude(r, ss);
        }
        if (n >= 64) return JO_ERR_BAD_STREAM;
        runs[n] = run; vals[n] = val; ++n;
        processed += 1 + run;
        if (processed > 63) return JO_ERR_BAD_STREAM; /* reference would overrun its array */
        if (run == 0 && val == 0) break;
    } while (processed != 63);
    if (r->eof) return JO_ERR_BAD_STREAM;
    zz[0] = (int16_t)(diff + *last_dc);
    *last_dc = zz[0];
    int k = 1;
    for (int i = 0; i + 1 < n; ++i) {       /* all but the last entry */
        for (int j = 0; j < runs[i]; ++j) zz[k++] = 0;
        zz[k++] = (int16_t)vals[i];
    }
    for (; k < 64; ++k) zz[k] = 0;
    return JO_OK;
}

/* Unstuffed scan bytes -> coefficient blocks ([mcu][ch][64], zigzag order).
 * *consumed_bytes = read position after byte alignment (encoder.cpp:63). */
int jo_scan_to_coefs(const uint8_t* scan, size_t len, size_t n_mcu, int16_t* coefs, size_t* consumed_bytes) {
    jo_init_dlut();
    jo_br r = {scan, len, 0, 0, 0};
    int16_t last[3] = {0, 0, 0};
    for (size_t m = 0; m < n_mcu; ++m)
        for (int ch = 0; ch < 3; ++ch) {
            int rc = jo_decode_block(&r, jo_is_lum(ch), &last[ch], coefs + (m * 3 + (size_t)ch) * 64);
            if (rc) return rc;
        }
    if (r.bit > 0) { r.bit = 0; r.byte++; }
    if (consumed_bytes) *consumed_bytes = r.byte;
    return JO_OK;
}

/* Coefficient blocks -> RGB image: entropy_encoder.cpp:79-92 (de-zigzag),
 * quantiser.cpp:72-85, discrete_cosine_transform.cpp:8-11,21-27,90-156,
 * colour_mapping.cpp:69-92, block_grid.cpp:87-99 (cropped store). */
int jo_coefs_to_image(const int16_t* coefs, int w, int h, const uint16_t lumq[64], const uint16_t chrq[64],
                      uint8_t* rgb) {
    if (w <= 0 || h <= 0) return JO_ERR_RANGE;
    int bw = (w + 7) / 8, bh = (h + 7) / 8;
    for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
            uint8_t ycc[3][64];
            for (int ch = 0; ch < 3; ++ch) {
                const int16_t* zz = coefs + ((size_t)(by * bw + bx) * 3 + (size_t)ch) * 64;
                int16_t nat[64];
                float deq[64];
                int8_t px[64];
                for (int k = 0; k < 64; ++k) nat[JO_ZIGZAG[k]] = zz[k];
                jo_dequantise(nat, jo_is_lum(ch) ? lumq : chrq, deq);
                jo_idct(deq, px);
                for (int i = 0; i < 64; ++i) ycc[ch][i] = (uint8_t)(px[i] + 128);
            }
            for (int row = 0; row < 8; ++row) {
                int yy = by * 8 + row; if (yy >= h) break;
                for (int col = 0; col < 8; ++col) {
                    int xx = bx * 8 + col; if (xx >= w) break;
                    int i = 8 * row + col;
                    uint8_t* px = rgb + 3 * ((size_t)yy * (size_t)w + (size_t)xx);
                    if (jo_mapper) { px[0] = ycc[0][i]; px[1] = ycc[1][i]; px[2] = ycc[2][i]; }
                    else jo_ycbcr_to_rgb_px(ycc[0][i], ycc[1][i], ycc[2][i], px);
                }
            }
        }
    return JO_OK;
}

/* Whole decode: encoder.cpp:44-71.  Output dimensions come from the caller
 * (the JPEGImage struct fields, :46), not from the SOF0 header; the quantiser
 * tables are those of `quality`, not the file's DQT (:187-188). */
int jo_decode(const uint8_t* jpeg, size_t len, int w, int h, int quality, uint8_t* rgb) {
    int hw = 0, hh = 0;
    if (jo_check_header(jpeg, len, &hw, &hh) != 0) return JO_ERR_BAD_HEADER;
    uint8_t* buf = (uint8_t*)malloc(len);
    if (!buf) return JO_ERR_CAPACITY;
    memcpy(buf, jpeg, len);
    size_t nlen = jo_remove_stuffing(buf, JO_HEADER_BYTES, len);
    size_t n_mcu = (size_t)((w + 7) / 8) * (size_t)((h + 7) / 8);
    int16_t* coefs = (int16_t*)malloc(n_mcu * 3 * 64 * sizeof(int16_t));
    if (!coefs) { free(buf); return JO_ERR_CAPACITY; }
    size_t used = 0;
    int rc = jo_scan_to_coefs(buf + JO_HEADER_BYTES, nlen - JO_HEADER_BYTES, n_mcu, coefs, &used);
    if (!rc) {
        size_t p = JO_HEADER_BYTES + used;
        if (p + 2 > nlen || buf[p] != 0xFF || buf[p + 1] != 0xD9) rc = JO_ERR_NO_EOI;
    }
    if (!rc) {
        uint16_t lumq[64], chrq[64];
        jo_quant_tables(quality, lumq, chrq);
        rc = jo_coefs_to_image(coefs, w, h, lumq, chrq, rgb);
    }
    free(coefs);
    free(buf);
    return rc;
}

/* jo_decode with the Huffman tables the file itself carries (files of jo_encode_optimised; the reference's
 * decoder skips the DHT segment, encoder.cpp:240, and always uses its own tables). */
int jo_decode_file_tables(const uint8_t* jpeg, size_t len, int w, int h, int quality, uint8_t* rgb) {
    if (len < JO_HEADER_BYTES) return JO_ERR_BAD_HEADER;
    jo_use_huffman_spec(jpeg + 177);
    int rc = jo_decode(jpeg, len, w, h, quality, rgb);
    jo_use_huffman_spec(NULL);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Synthetic inputs and hashing (shared by tests, bench and fixtures)         */
/* ------------------------------------------------------------------------- */

/* MT19937, the generator behind std::mt19937: the recorded reference hashes
 * were taken on images drawn from std::mt19937(1234). */
typedef struct { uint32_t mt[624]; int idx; } jo_mt;
static void jo_mt_seed(jo_mt* s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static uint32_t jo_mt_next(jo_mt* s) {
    if (s->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* mode 1 "noise": three generator draws per pixel, low byte each.
 * mode 0 "smooth": per-channel gradients plus noise 0..8 on r (added) and g
 * (subtracted).  Row-major, tightly packed RGB8. */
void jo_gen_image(int w, int h, int mode, uint32_t seed, uint8_t* rgb) {
    jo_mt s;
    jo_mt_seed(&s, seed);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int r, g, b;
            if (mode == 1) {
                r = (uint8_t)jo_mt_next(&s); g = (uint8_t)jo_mt_next(&s); b = (uint8_t)jo_mt_next(&s);
            } else {
                r = (x * 255) / (w > 1 ? w - 1 : 1);
                g = (y * 255) / (h > 1 ? h - 1 : 1);
                b = ((x + y) * 255) / (w + h);
                r = r + (int)(jo_mt_next(&s) % 9u); if (r > 255) r = 255;
                g = g - (int)(jo_mt_next(&s) % 9u); if (g < 0) g = 0;
            }
            uint8_t* p = rgb + 3 * ((size_t)y * (size_t)w + (size_t)x);
            p[0] = (uint8_t)r; p[1] = (uint8_t)g; p[2] = (uint8_t)b;
        }
}

/* FNV-1a, 64 bit.  `basis` is the offset basis: the standard one is
 * 0xcbf29ce484222325 (14695981039346656037); the hashes recorded from the
 * reference build used 1469598103934665603 (0x14650fb0739d0383, the standard
 * decimal constant with its last digit missing), so checking against them needs
 * that value. */
uint64_t jo_fnv1a64(const uint8_t* d, size_t n, uint64_t basis) {
    uint64_t h = basis;
    for (size_t i = 0; i < n; ++i) { h ^= d[i]; h *= 0x100000001b3ull; }
    return h;
}
