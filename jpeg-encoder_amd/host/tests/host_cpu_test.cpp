// Host-only checks of the container classes (no GPU call is made): BitStream bit
// packing / stuffing / reading, block grid iteration with edge replication and
// cropping, BMP file round trip.  Exit status 0 = all passed.
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "encoder.hpp"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

int main(int argc, char** argv) {
    using namespace jpeg;
    {   // MSB-first packing, partial byte visible through readByte/getSize, zero padding
        BitStream s;
        s.pushBitsu16(0b101, 3);
        CHECK(s.getSize() == 1 && s.committedBytes() == 0 && s.readByte(0) == 0b10100000);
        s.pushBitsu16(0x1FF, 9);            // 101 1 1111 1111 -> 0xBF 0xF.
        CHECK(s.committedBytes() == 1 && s.readByte(0) == 0xBF && s.readByte(1) == 0xF0);
        s.pushBitsu16(0xFFFF, 2);           // only the two low bits are taken
        s.pushIntoAlignment();
        CHECK(s.getSize() == 2 && s.readByte(1) == 0xFC);
        s.pushWord(0xFFD9);
        CHECK(s.getSize() == 4 && s.readByte(2) == 0xFF && s.readByte(3) == 0xD9);
        BitStreamReadProgress p;
        CHECK(s.readNextBit(p) && !s.readNextBit(p) && s.readNextBit(p) && p.currentBit == 3);
        CHECK(s.readNextAlignedByte(p) == 0xFC && p.currentByte == 2);
        CHECK(s.readNextAlignedWord(p) == 0xFFD9);
    }
    {   // stuffing touches committed bytes from `from` on; unstuffing spares the last two bytes
        BitStream s;
        const uint8_t raw[] = {0xFF, 0x12, 0xFF, 0xFF, 0x00, 0x34, 0xFF};
        for (uint8_t b : raw) s.pushByte(b);
        s.stuffBytes(1);
        const uint8_t want[] = {0xFF, 0x12, 0xFF, 0x00, 0xFF, 0x00, 0x00, 0x34, 0xFF, 0x00};
        CHECK(s.getSize() == sizeof(want) && std::memcmp(s.getDataPtr(), want, sizeof(want)) == 0);
        s.pushWord(0xFFD9);
        BitStreamReadProgress p;
        p.currentByte = 1;
        s.removeStuffedBytes(p);
        const uint8_t back[] = {0xFF, 0x12, 0xFF, 0xFF, 0x00, 0x34, 0xFF, 0xFF, 0xD9};
        CHECK(s.getSize() == sizeof(back) && std::memcmp(s.getDataPtr(), back, sizeof(back)) == 0);
    }
    {   // block grid: raster order, edge replication (block_grid.cpp:16-26), end() reached exactly
        for (auto [w, h] : {std::pair{13, 7}, {8, 8}, {1, 1}, {17, 33}, {24, 16}, {9, 20}}) {
            BitmapImageRGB bmp{static_cast<uint16_t>(w), static_cast<uint16_t>(h)};
            for (int i = 0; i < w * h; ++i) bmp.m_imageData[i] = {uint8_t(i), uint8_t(i >> 8), uint8_t(i * 7)};
            InputBlockGrid grid(bmp);
            int n = 0;
            const int bw = (w + 7) / 8, bh = (h + 7) / 8;
            for (auto it = grid.begin(); it != grid.end(); ++it, ++n) {
                const int bx = n % bw, by = n / bw;
                BlockGrid::Block b = *it;
                for (int r = 0; r < 8; ++r)
                    for (int c = 0; c < 8; ++c) {
                        const int yy = std::min(by * 8 + r, h - 1), xx = std::min(bx * 8 + c, w - 1);
                        const auto& want = bmp.m_imageData[size_t(yy) * w + xx];
                        const auto& got = b.m_blockPixelData[8 * r + c];
                        if (got.r != want.r || got.g != want.g || got.b != want.b) { CHECK(!"block pixel"); r = c = 8; }
                    }
                CHECK(it.isLastCol() == (bx == bw - 1) && it.isLastRow() == (by == bh - 1));
            }
            CHECK(n == bw * bh);
            // output grid: scatter the same blocks back, cropped
            OutputBlockGrid out{static_cast<uint16_t>(w), static_cast<uint16_t>(h)};
            for (auto it = grid.begin(); it != grid.end(); ++it) out.processNextBlock(*it);
            CHECK(out.atEnd());
            BitmapImageRGB copy = out.getBitmapRGB();
            CHECK(copy.m_width == w && copy.height == h &&
                  std::memcmp(copy.m_imageData.data(), bmp.m_imageData.data(), size_t(w) * h * 3) == 0);
        }
    }
    if (argc > 1) {  // BMP files: padding (width 37 -> 111-byte rows padded to 112), bottom-up storage
        std::string dir = argv[1];
        std::mt19937 rng(3);
        BitmapImageRGB bmp(37, 19);
        for (auto& p : bmp.m_imageData) p = {uint8_t(rng()), uint8_t(rng()), uint8_t(rng())};
        CHECK(bmp.saveToFile(dir + "/t.bmp"));
        BitmapImageRGB loaded(dir + "/t.bmp");
        CHECK(loaded.m_width == 37 && loaded.height == 19 && loaded.m_fileSize == 54 + 112 * 19);
        CHECK(std::memcmp(loaded.m_imageData.data(), bmp.m_imageData.data(), 37 * 19 * 3) == 0);
        BitmapImageRGB missing(dir + "/does-not-exist.bmp");
        CHECK(missing.m_width == 0 && missing.height == 0 && missing.m_imageData.empty());
    }
    {   // quantiser tables and header segments come from the library's host helpers (no GPU)
        Quantiser q75(75), q100(100), q0(-5);
        CHECK(q75.luminanceTable()[0] == 8 && q75.chrominanceTable()[63] == 50 && q100.luminanceTable()[17] == 1);
        CHECK(q0.luminanceTable() == Quantiser(1).luminanceTable());
        BitStream s;
        q75.encodeHeaderQuantisationTables(s);
        CHECK(s.getSize() == 134 && s.readByte(0) == 0xFF && s.readByte(1) == 0xDB && s.readByte(3) == 132 && s.readByte(5) == 8);
        BitStream d;
        HuffmanEncoder().encodeHeaderEntropyTables(d);
        CHECK(d.getSize() == 420 && d.readByte(1) == 0xC4 && d.readByte(4) == 0x00 && d.readByte(4 + 17 + 12) == 0x10);
    }
    std::printf(failures ? "%d check(s) failed\n" : "all host checks passed\n", failures);
    return failures ? 1 : 0;
}
