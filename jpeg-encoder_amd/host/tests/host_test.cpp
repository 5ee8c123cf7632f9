// Exercises the C++ host interface the way callers of the reference use it
// (examples/encode-decode.cpp:102-118 of the reference): BaselineEncoder(q),
// encode(bitmap, jpeg), decode(jpeg, bitmap) -- and prints hashes that the Python
// GPU test compares with the oracle's.  Also drives the per-block component
// methods through the reference's own loop structure (encoder.cpp:14-42) and
// checks that they give the same file as the fused path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "encoder.hpp"

static uint64_t fnv(const uint8_t* d, size_t n, uint64_t basis) {
    uint64_t h = basis;
    for (size_t i = 0; i < n; ++i) { h ^= d[i]; h *= 0x100000001b3ull; }
    return h;
}

// Same synthetic images as oracle/jpeg_oracle.c jo_gen_image (std::mt19937 stream).
static jpeg::BitmapImageRGB make_image(int w, int h, int mode, unsigned seed) {
    jpeg::BitmapImageRGB bmp(static_cast<uint16_t>(w), static_cast<uint16_t>(h));
    std::mt19937 rng(seed);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int r, g, b;
            if (mode == 1) {
                r = uint8_t(rng()); g = uint8_t(rng()); b = uint8_t(rng());
            } else {
                r = (x * 255) / (w > 1 ? w - 1 : 1);
                g = (y * 255) / (h > 1 ? h - 1 : 1);
                b = ((x + y) * 255) / (w + h);
                r = std::min(255, r + int(rng() % 9));
                g = std::max(0, g - int(rng() % 9));
            }
            bmp.m_imageData[size_t(y) * w + x] = {uint8_t(r), uint8_t(g), uint8_t(b)};
        }
    return bmp;
}

// The reference's per-block encode loop, written against the component interfaces only.
static void encode_block_by_block(jpeg::BitmapImageRGB const& bmp, int quality, jpeg::BitStream& out,
                                  jpeg::ColourMapper const& mapper) {
    jpeg::SeparatedDiscreteCosineTransformer dct;
    jpeg::Quantiser quantiser(quality);
    jpeg::HuffmanEncoder huffman;
    out.clearStream();
    out.pushWord(jpeg::markerStartOfImageSegmentSOI);
    out.pushWord(jpeg::markerJFIFImageSegmentAPP0);
    out.pushWord(16);
    for (char c : {'J', 'F', 'I', 'F'}) out.pushByte(uint8_t(c));
    out.pushByte(0); out.pushWord(0x0102); out.pushByte(0); out.pushWord(0x0010); out.pushWord(0x0010); out.pushWord(0);
    quantiser.encodeHeaderQuantisationTables(out);
    out.pushWord(jpeg::markerStartOfFrame0SOF0);
    out.pushWord(17); out.pushByte(8); out.pushWord(bmp.height); out.pushWord(bmp.m_width); out.pushByte(3);
    out.pushByte(1); out.pushByte(0x11); out.pushByte(0);
    out.pushByte(2); out.pushByte(0x11); out.pushByte(1);
    out.pushByte(3); out.pushByte(0x11); out.pushByte(1);
    huffman.encodeHeaderEntropyTables(out);
    out.pushWord(jpeg::markerStartOfScanSegmentSOS);
    out.pushWord(12); out.pushByte(3);
    out.pushByte(1); out.pushByte(0x00); out.pushByte(2); out.pushByte(0x11); out.pushByte(3); out.pushByte(0x11);
    out.pushWord(0x003F); out.pushByte(0);
    const size_t scanStart = out.getSize();
    jpeg::InputBlockGrid grid(bmp);
    int16_t lastDC[3] = {0, 0, 0};
    for (auto it = grid.begin(); it != grid.end(); ++it) {
        jpeg::ColourMappedBlockData mapped = mapper.map(*it);
        for (uint8_t ch = 0; ch < 3; ++ch) {
            auto dctData = dct.transform(mapped.m_data[ch]);
            auto q = quantiser.quantise(dctData, mapper.isLuminanceComponent(ch));
            huffman.encode(q, lastDC[ch], out, mapper.isLuminanceComponent(ch));
        }
    }
    out.stuffBytes(scanStart);
    out.pushIntoAlignment();
    out.pushWord(jpeg::markerEndOfImageSegmentEOI);
}

// The reference's per-block decode loop (encoder.cpp:44-71) on the component interfaces.
static jpeg::BitmapImageRGB decode_block_by_block(jpeg::JPEGImage jpg, int quality, jpeg::ColourMapper const& mapper) {
    jpeg::SeparatedDiscreteCosineTransformer dct;
    jpeg::Quantiser quantiser(quality);
    jpeg::HuffmanEncoder huffman;
    jpeg::OutputBlockGrid grid(jpg.m_width, jpg.m_height);
    jpeg::BitStreamReadProgress progress;
    progress.currentByte = 607;
    jpg.m_compressedImageData.removeStuffedBytes(progress);
    int16_t lastDC[3] = {0, 0, 0};
    while (!grid.atEnd()) {
        jpeg::ColourMappedBlockData block;
        for (uint8_t ch = 0; ch < 3; ++ch) {
            auto q = huffman.decode(jpg.m_compressedImageData, progress, lastDC[ch], mapper.isLuminanceComponent(ch));
            block.m_data[ch] = dct.inverseTransform(quantiser.dequantise(q, mapper.isLuminanceComponent(ch)));
        }
        grid.processNextBlock(mapper.unmap(block));
    }
    if (jpg.m_compressedImageData.readNextAlignedWord(progress) != jpeg::markerEndOfImageSegmentEOI)
        std::printf("blockwise_decode_eoi_missing\n");
    return grid.getBitmapRGB();
}

int main(int argc, char** argv) {
    std::setvbuf(stdout, nullptr, _IOLBF, 0);  // lines survive an abort
    const uint64_t basis = 0x14650fb0739d0383ull;
    struct Case { int w, h, q, mode; };
    const Case cases[] = {{512, 512, 75, 0}, {512, 512, 75, 1}, {13, 7, 75, 1}, {17, 33, 90, 1}, {8, 8, 50, 0},
                          {1, 1, 75, 1}, {1920, 1080, 90, 0}};
    for (const Case& c : cases) {
        jpeg::BitmapImageRGB bmp = make_image(c.w, c.h, c.mode, 1234);
        jpeg::BaselineEncoder encoder(c.q);
        jpeg::JPEGImage jpg;
        encoder.encode(bmp, jpg);
        jpeg::BitmapImageRGB dec;
        encoder.decode(jpg, dec);
        std::printf("case %d %d %d %d bytes %zu jpeg %016llx rgb %016llx saving %d dims %u %u\n", c.w, c.h, c.q, c.mode,
                    size_t(jpg.m_fileSize),
                    (unsigned long long)fnv(jpg.m_compressedImageData.getDataPtr(), jpg.m_compressedImageData.getSize(), basis),
                    (unsigned long long)fnv(reinterpret_cast<const uint8_t*>(dec.m_imageData.data()), dec.m_imageData.size() * 3, basis),
                    int(jpg.m_supportsSaving), unsigned(dec.m_width), unsigned(dec.height));
    }
    {   // per-block component methods against the fused path
        jpeg::BitmapImageRGB bmp = make_image(43, 21, 1, 77);
        jpeg::BaselineEncoder encoder(80);
        jpeg::JPEGImage jpg;
        encoder.encode(bmp, jpg);
        jpeg::BitStream blockwise;
        encode_block_by_block(bmp, 80, blockwise, jpeg::RGBToYCbCrMapper());
        const bool same = blockwise.getSize() == jpg.m_compressedImageData.getSize() &&
                          std::memcmp(blockwise.getDataPtr(), jpg.m_compressedImageData.getDataPtr(), blockwise.getSize()) == 0;
        std::printf("blockwise_encode_matches %d bytes %zu\n", int(same), blockwise.getSize());
        jpeg::BitmapImageRGB fused;
        encoder.decode(jpg, fused);
        jpeg::BitmapImageRGB perBlock = decode_block_by_block(jpg, 80, jpeg::RGBToYCbCrMapper());
        const bool sameDec = fused.m_imageData.size() == perBlock.m_imageData.size() &&
                             std::memcmp(fused.m_imageData.data(), perBlock.m_imageData.data(), fused.m_imageData.size() * 3) == 0;
        std::printf("blockwise_decode_matches %d\n", int(sameDec));
    }
    {   // the other mapper the reference ships (colour_mapping.cpp:16-42), injected: fused path == per-block loop
        class PassThrough final : public jpeg::Encoder {
        public:
            PassThrough() : Encoder(std::make_unique<jpeg::RGBToRGBMapper>(), std::make_unique<jpeg::SeparatedDiscreteCosineTransformer>(),
                                    std::make_unique<jpeg::Quantiser>(50), std::make_unique<jpeg::HuffmanEncoder>()) {}
        private:
            bool supportsSaving() const override { return false; }
        } passThrough;
        jpeg::BitmapImageRGB bmp = make_image(40, 24, 1, 1);
        jpeg::JPEGImage jpg;
        passThrough.encode(bmp, jpg);
        jpeg::BitStream blockwise;
        encode_block_by_block(bmp, 50, blockwise, jpeg::RGBToRGBMapper());
        const bool same = blockwise.getSize() == jpg.m_compressedImageData.getSize() &&
                          std::memcmp(blockwise.getDataPtr(), jpg.m_compressedImageData.getDataPtr(), blockwise.getSize()) == 0;
        std::printf("passthrough_encode_matches %d bytes %zu\n", int(same), blockwise.getSize());
        jpeg::BitmapImageRGB fused;
        passThrough.decode(jpg, fused);
        jpeg::BitmapImageRGB perBlock = decode_block_by_block(jpg, 50, jpeg::RGBToRGBMapper());
        const bool sameDec = fused.m_imageData.size() == perBlock.m_imageData.size() && !fused.m_imageData.empty() &&
                             std::memcmp(fused.m_imageData.data(), perBlock.m_imageData.data(), fused.m_imageData.size() * 3) == 0;
        std::printf("passthrough_decode_matches %d\n", int(sameDec));
    }
    {   // a user's colour mapper (a subclass the library knows nothing about): the encoder falls back to the
        // reference's loop over blocks and calls the user's virtuals; this one restates the pass-through
        // mapping, so its files must equal those of the library's RGBToRGBMapper on the fused path
        class UserMapper final : public jpeg::ColourMapper {
        protected:
            jpeg::ColourMappedBlockData applyMapping(jpeg::BlockGrid::Block const& in) const override {
                jpeg::ColourMappedBlockData out;
                for (size_t i = 0; i < in.m_blockPixelData.size(); ++i) {
                    out.m_data[0][i] = in.m_blockPixelData[i].r;
                    out.m_data[1][i] = in.m_blockPixelData[i].g;
                    out.m_data[2][i] = in.m_blockPixelData[i].b;
                }
                ++calls;
                return out;
            }
            jpeg::BlockGrid::Block reverseMapping(jpeg::ColourMappedBlockData const& in) const override {
                jpeg::BlockGrid::Block out;
                for (size_t i = 0; i < out.m_blockPixelData.size(); ++i)
                    out.m_blockPixelData[i] = {in.m_data[0][i], in.m_data[1][i], in.m_data[2][i]};
                return out;
            }
            bool componentIsLuminance(uint8_t) const override { return true; }
        public:
            mutable size_t calls = 0;
        };
        class MyNewEncoder final : public jpeg::Encoder {   // the shape of the reference's readme example
        public:
            MyNewEncoder(int quality, std::unique_ptr<jpeg::ColourMapper> mapper)
                : Encoder(std::move(mapper), std::make_unique<jpeg::SeparatedDiscreteCosineTransformer>(),
                          std::make_unique<jpeg::Quantiser>(quality), std::make_unique<jpeg::HuffmanEncoder>()) {}
        private:
            bool supportsSaving() const override { return false; }
        };
        auto userMapper = std::make_unique<UserMapper>();
        UserMapper const* counter = userMapper.get();
        MyNewEncoder user(65, std::move(userMapper)), library(65, std::make_unique<jpeg::RGBToRGBMapper>());
        jpeg::BitmapImageRGB bmp = make_image(45, 30, 1, 3);
        jpeg::JPEGImage a, b;
        user.encode(bmp, a);
        library.encode(bmp, b);
        const bool same = a.m_compressedImageData.getSize() == b.m_compressedImageData.getSize() && a.m_fileSize == b.m_fileSize &&
                          std::memcmp(a.m_compressedImageData.getDataPtr(), b.m_compressedImageData.getDataPtr(), a.m_compressedImageData.getSize()) == 0;
        std::printf("user_mapper_encode_matches %d blocks_mapped %zu saving %d jpeg %016llx\n", int(same), counter->calls, int(a.m_supportsSaving),
                    (unsigned long long)fnv(a.m_compressedImageData.getDataPtr(), a.m_compressedImageData.getSize(), basis));
        jpeg::BitmapImageRGB da, db;
        user.decode(a, da);
        library.decode(b, db);
        const bool sameDec = da.m_imageData.size() == db.m_imageData.size() && da.m_width == 45 && da.height == 30 &&
                             std::memcmp(da.m_imageData.data(), db.m_imageData.data(), da.m_imageData.size() * 3) == 0;
        std::printf("user_mapper_decode_matches %d rgb %016llx\n", int(sameDec),
                    (unsigned long long)fnv(reinterpret_cast<const uint8_t*>(da.m_imageData.data()), da.m_imageData.size() * 3, basis));
    }
    {   // a subclass of HuffmanEncoder written against the reference's interface (entropy_encoder.hpp:40-51): it
        // overrides the protected final-encoding step; encode / decode of the base class must route through it
        class CountingHuffman final : public jpeg::HuffmanEncoder {
        public:
            mutable size_t encoded = 0, decoded = 0, entries = 0;
        protected:
            void applyFinalEncoding(jpeg::RunLengthEncodedBlockChannelData const& in, jpeg::BitStream& out, bool lum) const override {
                ++encoded;
                entries += in.m_acCoefficients.size();
                HuffmanEncoder::applyFinalEncoding(in, out, lum);
            }
            jpeg::RunLengthEncodedBlockChannelData removeFinalEncoding(jpeg::BitStream const& in, jpeg::BitStreamReadProgress& p, bool lum) const override {
                ++decoded;
                return HuffmanEncoder::removeFinalEncoding(in, p, lum);
            }
        };
        class WithCounting final : public jpeg::Encoder {
        public:
            WithCounting(std::unique_ptr<jpeg::EntropyEncoder> e)
                : Encoder(std::make_unique<jpeg::RGBToYCbCrMapper>(), std::make_unique<jpeg::SeparatedDiscreteCosineTransformer>(),
                          std::make_unique<jpeg::Quantiser>(75), std::move(e)) {}
        private:
            bool supportsSaving() const override { return true; }
        };
        auto coder = std::make_unique<CountingHuffman>();
        CountingHuffman const* counter = coder.get();
        WithCounting custom(std::move(coder));
        jpeg::BaselineEncoder baseline(75);
        jpeg::BitmapImageRGB bmp = make_image(29, 18, 0, 8);
        jpeg::JPEGImage a, b;
        custom.encode(bmp, a);
        baseline.encode(bmp, b);
        const bool same = a.m_compressedImageData.getSize() == b.m_compressedImageData.getSize() &&
                          std::memcmp(a.m_compressedImageData.getDataPtr(), b.m_compressedImageData.getDataPtr(), a.m_compressedImageData.getSize()) == 0;
        jpeg::BitmapImageRGB da, db;
        custom.decode(a, da);
        baseline.decode(b, db);
        const bool sameDec = da.m_imageData.size() == db.m_imageData.size() && !da.m_imageData.empty() &&
                             std::memcmp(da.m_imageData.data(), db.m_imageData.data(), da.m_imageData.size() * 3) == 0;
        std::printf("user_entropy_coder encode_matches %d decode_matches %d encoded %zu decoded %zu\n", int(same), int(sameDec),
                    counter->encoded, counter->decoded);
        // the final-encoding step takes the entries as they are given: an entry that no code exists for is the
        // reference's runtime error (entropy_encoder.cpp:748), printed and swallowed by encode
        class BadEntries final : public jpeg::HuffmanEncoder {
        protected:
            void applyFinalEncoding(jpeg::RunLengthEncodedBlockChannelData const& in, jpeg::BitStream& out, bool lum) const override {
                jpeg::RunLengthEncodedBlockChannelData bad = in;
                bad.m_acCoefficients.insert(bad.m_acCoefficients.begin(), {3, 0});
                HuffmanEncoder::applyFinalEncoding(bad, out, lum);
            }
        };
        WithCounting broken(std::make_unique<BadEntries>());
        jpeg::JPEGImage c;
        broken.encode(bmp, c);
        std::printf("bad_entries_done\n");
        // the reference narrows an entry's run to 8 bits before it looks at it (entropy_encoder.cpp:733): a run of 256
        // with value 0 is the end-of-block entry, and a file written with such entries is the baseline's file
        class WideRuns final : public jpeg::HuffmanEncoder {
        protected:
            void applyFinalEncoding(jpeg::RunLengthEncodedBlockChannelData const& in, jpeg::BitStream& out, bool lum) const override {
                jpeg::RunLengthEncodedBlockChannelData wide = in;
                for (auto& e : wide.m_acCoefficients)
                    if (e.m_value == 0 && e.m_runLength == 0) e.m_runLength = 256;
                HuffmanEncoder::applyFinalEncoding(wide, out, lum);
            }
        };
        WithCounting widened(std::make_unique<WideRuns>());
        jpeg::JPEGImage d;
        widened.encode(bmp, d);
        const bool sameWide = d.m_compressedImageData.getSize() == b.m_compressedImageData.getSize() &&
                              std::memcmp(d.m_compressedImageData.getDataPtr(), b.m_compressedImageData.getDataPtr(), d.m_compressedImageData.getSize()) == 0;
        std::printf("wide_runs_narrowed encode_matches %d\n", int(sameWide));
    }
    {   // the injected coder with per-image tables: a smaller file that decodes to the same pixels; mixed with a
        // user's component it cannot work (two passes do not fit the per-block loop) and says so
        class WithTables final : public jpeg::Encoder {
        public:
            explicit WithTables(std::unique_ptr<jpeg::ColourMapper> m)
                : Encoder(std::move(m), std::make_unique<jpeg::SeparatedDiscreteCosineTransformer>(),
                          std::make_unique<jpeg::Quantiser>(60), std::make_unique<jpeg::OptimisedHuffmanEncoder>()) {}
        private:
            bool supportsSaving() const override { return true; }
        };
        WithTables tailored(std::make_unique<jpeg::RGBToYCbCrMapper>());
        jpeg::BaselineEncoder baseline(60);
        jpeg::BitmapImageRGB bmp = make_image(160, 96, 0, 12);
        jpeg::JPEGImage a, b;
        tailored.encode(bmp, a);
        baseline.encode(bmp, b);
        jpeg::BitmapImageRGB da, db;
        tailored.decode(a, da);
        baseline.decode(b, db);
        const bool sameDec = da.m_imageData.size() == db.m_imageData.size() && !da.m_imageData.empty() &&
                             std::memcmp(da.m_imageData.data(), db.m_imageData.data(), da.m_imageData.size() * 3) == 0;
        std::printf("tailored_tables bytes %zu baseline %zu decode_matches %d jpeg_fnv %016llx\n", a.m_compressedImageData.getSize(),
                    b.m_compressedImageData.getSize(), int(sameDec),
                    (unsigned long long)fnv(a.m_compressedImageData.getDataPtr(), a.m_compressedImageData.getSize(), basis));
        class Identity final : public jpeg::RGBToYCbCrMapper {};
        WithTables mixed(std::make_unique<Identity>());
        jpeg::JPEGImage c;
        mixed.encode(bmp, c);
        std::printf("tailored_tables_mixed bytes %zu\n", c.m_compressedImageData.getSize());
    }
    {   // wall time of the two calls a user of the classes makes, on a 3840x2160 picture
        jpeg::BitmapImageRGB bmp = make_image(3840, 2160, 0, 1234);
        jpeg::BaselineEncoder encoder(75);
        jpeg::JPEGImage jpg;
        jpeg::BitmapImageRGB dec;
        for (int it = 0; it < 4; ++it) {
            const auto t0 = std::chrono::steady_clock::now();
            encoder.encode(bmp, jpg);
            const auto t1 = std::chrono::steady_clock::now();
            encoder.decode(jpg, dec);
            const auto t2 = std::chrono::steady_clock::now();
            std::printf("timing_4k encode_ms %.2f decode_ms %.2f\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    }
    {   // damaged header: the reference's message on stdout, output untouched
        jpeg::BitmapImageRGB bmp = make_image(16, 16, 0, 1);
        jpeg::BaselineEncoder encoder(75);
        jpeg::JPEGImage jpg;
        encoder.encode(bmp, jpg);
        jpeg::JPEGImage broken;
        broken.m_width = 16; broken.m_height = 16;
        std::vector<uint8_t> bytes(jpg.m_compressedImageData.getDataPtr(), jpg.m_compressedImageData.getDataPtr() + jpg.m_compressedImageData.getSize());
        bytes[1] = 0;
        broken.m_compressedImageData.appendBytes(bytes.data(), bytes.size());
        jpeg::BitmapImageRGB dec;
        encoder.decode(broken, dec);
        std::printf("broken_header_pixels %zu\n", dec.m_imageData.size());
    }
    if (argc > 1) {  // BMP round trip through files
        std::string dir = argv[1];
        jpeg::BitmapImageRGB bmp = make_image(37, 19, 1, 5);
        bmp.saveToFile(dir + "/in.bmp");
        jpeg::BitmapImageRGB loaded(dir + "/in.bmp");
        const bool same = loaded.m_width == 37 && loaded.height == 19 &&
                          std::memcmp(loaded.m_imageData.data(), bmp.m_imageData.data(), 37 * 19 * 3) == 0;
        std::printf("bmp_round_trip %d\n", int(same));
    }
    return 0;
}
