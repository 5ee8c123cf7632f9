// Host classes of namespace jpeg (see ../inc/jpeg_amd.hpp).  Image arithmetic is
// delegated to the C-ABI in include/jpegenc_amd.h; what remains here is container
// plumbing (bit stream, block iterators, BMP files) and the orchestration of
// Encoder::encode / ::decode with the reference's error convention.
#include "jpeg_amd.hpp"

#include <cstring>
#include <fstream>
#include <iostream>
#include <atomic>
#include <mutex>
#include <typeinfo>
#include <stdexcept>

#include "../../../include/jpegenc_amd.h"

namespace jpeg {

namespace {

// One context shared by the per-block component methods (they carry no state of
// their own in the reference, so they cannot own one).
std::mutex g_component_mutex;
jea_ctx* g_component_ctx = nullptr;

jea_ctx* componentContext() {
    if (!g_component_ctx) {
        if (jea_create(0, &g_component_ctx) != JEA_OK)
            throw std::runtime_error(std::string("GPU context: ") + jea_last_error(nullptr));
    }
    return g_component_ctx;
}

void check(jea_ctx* ctx, int rc) {
    if (rc != JEA_OK) throw std::runtime_error(jea_last_error(ctx));
}

constexpr uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

uint32_t le32(uint8_t const* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(uint8_t const* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

// ---- BitmapImageRGB -------------------------------------------------------------

BitmapImageRGB::BitmapImageRGB() : m_width{0}, height{0}, m_fileSize{0} {}

BitmapImageRGB::BitmapImageRGB(uint16_t w, uint16_t h) : m_width{w}, height{h}, m_fileSize{0} {
    m_imageData.resize(static_cast<size_t>(w) * h);
}

BitmapImageRGB::BitmapImageRGB(std::string const& loadPath) : m_width{0}, height{0}, m_fileSize{0} {
    std::ifstream file(loadPath, std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    if (!file.good() && bytes.empty()) {
        std::cout << "Failed to load image at " << loadPath << " (cannot read file)\n";
        return;
    }
    if (!parse(bytes.data(), bytes.size())) {
        std::cout << "Failed to load image at " << loadPath << " (not an uncompressed 24/32-bit BMP)\n";
        return;
    }
    m_fileSize = bytes.size();
}

BitmapImageRGB::BitmapImageRGB(uint8_t const* buffer, int len) : m_width{0}, height{0}, m_fileSize{0} {
    if (!buffer || len <= 0 || !parse(buffer, static_cast<size_t>(len))) {
        std::cout << "Failed to load image (not an uncompressed 24/32-bit BMP)\n";
        return;
    }
    m_fileSize = static_cast<std::uintmax_t>(len);
}

// BITMAPFILEHEADER + BITMAPINFOHEADER (or later), BI_RGB, 24 or 32 bits per pixel;
// rows padded to four bytes, bottom-up unless the height is negative, BGR(A).
bool BitmapImageRGB::parse(uint8_t const* b, size_t len) {
    if (len < 54 || b[0] != 'B' || b[1] != 'M') return false;
    const uint32_t dataOffset = le32(b + 10), headerSize = le32(b + 14);
    if (headerSize < 40) return false;
    const int32_t w = static_cast<int32_t>(le32(b + 18)), hSigned = static_cast<int32_t>(le32(b + 22));
    const uint16_t bpp = le16(b + 28);
    const uint32_t compression = le32(b + 30);
    if (w <= 0 || hSigned == 0 || w > 65535 || (bpp != 24 && bpp != 32) || (compression != 0 && compression != 3))
        return false;
    const bool topDown = hSigned < 0;
    const uint32_t h = static_cast<uint32_t>(topDown ? -hSigned : hSigned);
    if (h > 65535) return false;
    const size_t bytesPerPixel = bpp / 8;
    const size_t stride = (static_cast<size_t>(w) * bytesPerPixel + 3) & ~size_t(3);
    if (dataOffset + stride * h > len) return false;
    m_width = static_cast<uint16_t>(w);
    height = static_cast<uint16_t>(h);
    m_imageData.resize(static_cast<size_t>(w) * h);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t const* row = b + dataOffset + stride * (topDown ? y : h - 1 - y);
        for (int32_t x = 0; x < w; ++x) {
            uint8_t const* px = row + bytesPerPixel * x;
            m_imageData[static_cast<size_t>(y) * w + x] = PixelData{px[2], px[1], px[0]};
        }
    }
    return true;
}

bool BitmapImageRGB::saveToFile(std::string const& path) const {
    const size_t stride = (static_cast<size_t>(m_width) * 3 + 3) & ~size_t(3);
    const uint32_t dataSize = static_cast<uint32_t>(stride * height), fileSize = 54 + dataSize;
    std::vector<uint8_t> out(fileSize, 0);
    auto put32 = [&](size_t at, uint32_t v) { for (int i = 0; i < 4; ++i) out[at + i] = static_cast<uint8_t>(v >> (8 * i)); };
    out[0] = 'B'; out[1] = 'M';
    put32(2, fileSize); put32(10, 54); put32(14, 40); put32(18, m_width); put32(22, height);
    out[26] = 1; out[28] = 24; put32(34, dataSize); put32(38, 2835); put32(42, 2835);
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* row = out.data() + 54 + stride * (height - 1u - y);
        for (uint32_t x = 0; x < m_width; ++x) {
            PixelData const& p = m_imageData[static_cast<size_t>(y) * m_width + x];
            row[3 * x] = p.b; row[3 * x + 1] = p.g; row[3 * x + 2] = p.r;
        }
    }
    std::ofstream file(path, std::ios::binary | std::ios::trunc);
    file.write(reinterpret_cast<char const*>(out.data()), static_cast<std::streamsize>(out.size()));
    return file.good();
}

// ---- BitStream --------------------------------------------------------------------

BitStreamReadProgress::BitStreamReadProgress() { reset(); }
void BitStreamReadProgress::reset() { currentByte = 0; currentBit = 0; }
void BitStreamReadProgress::advanceBits(size_t numBits) {
    const size_t total = currentBit + numBits;
    currentByte += total / 8;
    currentBit = total % 8;
}
void BitStreamReadProgress::advanceBit() { advanceBits(1); }
void BitStreamReadProgress::advanceIntoAlignment() {
    if (currentBit != 0) { currentBit = 0; ++currentByte; }
}

BitStream::BitStream() { clearStream(); }

// Revisions are unique across all streams of the process, so equal revisions mean equal contents
// (copies keep the number of the stream they were copied from).
static uint64_t nextStreamRevision() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

void BitStream::clearStream() {
    m_buffer = 0;
    m_bitsInBuffer = 0;
    m_stream.clear();
    m_revision = nextStreamRevision();
}

size_t BitStream::getSize() { return m_stream.size() + (m_bitsInBuffer > 0 ? 1 : 0); }

// Appends the `n` low bits of `data`, most significant first; a byte moves to the
// stream as soon as it is complete (semantics of bit_stream.cpp:44-67).
void BitStream::pushBitsu8(uint8_t data, size_t n) {
    m_revision = nextStreamRevision();
    for (size_t i = n; i-- > 0;) {
        m_buffer = static_cast<uint8_t>((m_buffer << 1) | ((data >> i) & 1u));
        if (++m_bitsInBuffer == 8) {
            m_stream.push_back(m_buffer);
            m_buffer = 0;
            m_bitsInBuffer = 0;
        }
    }
}

void BitStream::pushBitsu16(uint16_t data, size_t n) {
    if (n > 8) {
        pushBitsu8(static_cast<uint8_t>(data >> 8), n - 8);
        pushBitsu8(static_cast<uint8_t>(data), 8);
    } else {
        pushBitsu8(static_cast<uint8_t>(data), n);
    }
}

// The pending partial byte is kept left-aligned when read back, as the reference
// keeps it in m_buffer (bit_stream.cpp:83-91).
uint8_t BitStream::readByte(size_t byte) const {
    if (byte == m_stream.size()) return static_cast<uint8_t>(m_buffer << ((8 - m_bitsInBuffer) & 7));
    return m_stream[byte];
}

void BitStream::pushByte(uint8_t data) { pushBitsu8(data, 8); }
void BitStream::pushWord(uint16_t data) { pushBitsu16(data, 16); }
void BitStream::pushIntoAlignment() {
    if (m_bitsInBuffer > 0) pushBitsu8(0, 8 - m_bitsInBuffer);
}

bool BitStream::readNextBit(BitStreamReadProgress& progress) const {
    const bool bit = (readByte(progress.currentByte) >> (7 - progress.currentBit)) & 1u;
    progress.advanceBit();
    return bit;
}
uint8_t BitStream::readNextAlignedByte(BitStreamReadProgress& progress) const {
    progress.advanceIntoAlignment();
    return readByte(progress.currentByte++);
}
uint16_t BitStream::readNextAlignedWord(BitStreamReadProgress& progress) const {
    const uint16_t hi = readNextAlignedByte(progress);
    return static_cast<uint16_t>((hi << 8) | readNextAlignedByte(progress));
}
uint8_t const* BitStream::getDataPtr() const { return m_stream.data(); }

// A zero byte after every committed 0xFF from `from` on (intended semantics of
// bit_stream.cpp:127-134), built in one pass.
void BitStream::stuffBytes(size_t from) {
    m_revision = nextStreamRevision();
    std::vector<uint8_t> out;
    out.reserve(m_stream.size() + m_stream.size() / 64 + 16);
    out.insert(out.end(), m_stream.begin(), m_stream.begin() + static_cast<std::ptrdiff_t>(std::min(from, m_stream.size())));
    for (size_t i = from; i < m_stream.size(); ++i) {
        out.push_back(m_stream[i]);
        if (m_stream[i] == 0xFF) out.push_back(0x00);
    }
    m_stream.swap(out);
}

// Drops the zero byte after each 0xFF between the read position and the final two
// bytes (bit_stream.cpp:136-147).
void BitStream::removeStuffedBytes(BitStreamReadProgress const& progress) {
    m_revision = nextStreamRevision();
    if (m_stream.size() < progress.currentByte + 2) return;
    const size_t end = m_stream.size() - 2;
    size_t w = progress.currentByte;
    size_t r = progress.currentByte;
    while (r < end) {
        const uint8_t b = m_stream[r++];
        m_stream[w++] = b;
        if (b == 0xFF && r < m_stream.size()) {
            if (m_stream[r] == 0x00) ++r;
            else if (r < end) m_stream[w++] = m_stream[r++];
        }
    }
    while (r < m_stream.size()) m_stream[w++] = m_stream[r++];
    m_stream.resize(w);
}

void BitStream::appendBytes(uint8_t const* data, size_t len) {
    if (m_bitsInBuffer != 0) throw std::logic_error("BitStream::appendBytes on an unaligned stream");
    m_revision = nextStreamRevision();
    m_stream.insert(m_stream.end(), data, data + len);
}

void JPEGImage::saveToFile(std::string const& path) {
    if (!m_supportsSaving) {
        std::cerr << "Saving is not permitted for this configuration. Try using the baseline encoder.\n";
        return;
    }
    std::ofstream file(path.c_str(), std::ios::out | std::ios::binary | std::ios::app);
    file.write(reinterpret_cast<char const*>(m_compressedImageData.getDataPtr()),
               static_cast<std::streamsize>(m_compressedImageData.getSize()));
}

// ---- block grid ------------------------------------------------------------------
// Raster walk over 8x8 blocks (block_grid.cpp:8-82).  The iterator's pointer is
// the block's upper-left pixel; end() is one past the last pixel of the image.

InputBlockGrid::InputBlockGrid(BitmapImageRGB const& input) : m_imageData{input} {}

InputBlockGrid::BlockIterator::BlockIterator(underlying_pointer p, uint16_t w, uint16_t h)
    : m_ptr{p}, m_blockRowPos{0}, m_blockColPos{0}, m_gridWidth{w}, m_gridHeight{h} {}

bool InputBlockGrid::BlockIterator::isLastCol() const { return m_blockColPos + 1 == (m_gridWidth + 7) / 8; }
bool InputBlockGrid::BlockIterator::isLastRow() const { return m_blockRowPos + 1 == (m_gridHeight + 7) / 8; }
InputBlockGrid::BlockIterator::underlying_pointer InputBlockGrid::BlockIterator::getDataPtr() const { return m_ptr; }

// Copies the block; pixels right of / below the image repeat the last valid
// column / row (block_grid.cpp:16-26).
InputBlockGrid::BlockIterator::value_type InputBlockGrid::BlockIterator::operator*() const {
    Block out{};
    const int validCols = std::min<int>(8, m_gridWidth - 8 * m_blockColPos);
    const int validRows = std::min<int>(8, m_gridHeight - 8 * m_blockRowPos);
    for (int row = 0; row < 8; ++row) {
        const int srcRow = std::min(row, validRows - 1);
        for (int col = 0; col < 8; ++col) {
            const int srcCol = std::min(col, validCols - 1);
            out.m_blockPixelData[8 * row + col] = m_ptr[static_cast<size_t>(srcRow) * m_gridWidth + srcCol];
        }
    }
    return out;
}

InputBlockGrid::BlockIterator& InputBlockGrid::BlockIterator::operator++() {
    const int validCols = std::min<int>(8, m_gridWidth - 8 * m_blockColPos);
    if (!isLastCol()) {
        m_ptr += 8;
        ++m_blockColPos;
        return *this;
    }
    const int validRows = std::min<int>(8, m_gridHeight - 8 * m_blockRowPos);
    // from the last block of a block-row to the first block of the next one:
    // finish this pixel row, then skip the block's remaining pixel rows
    m_ptr += validCols + static_cast<size_t>(validRows - 1) * m_gridWidth;
    m_blockColPos = 0;
    ++m_blockRowPos;
    return *this;
}

InputBlockGrid::BlockIterator InputBlockGrid::BlockIterator::operator++(int) {
    BlockIterator before = *this;
    ++(*this);
    return before;
}

InputBlockGrid::BlockIterator InputBlockGrid::begin() const {
    return BlockIterator(m_imageData.m_imageData.data(), m_imageData.m_width, m_imageData.height);
}
InputBlockGrid::BlockIterator InputBlockGrid::end() const {
    return BlockIterator(m_imageData.m_imageData.data() + static_cast<size_t>(m_imageData.m_width) * m_imageData.height,
                         m_imageData.m_width, m_imageData.height);
}

OutputBlockGrid::OutputBlockGrid(uint16_t width, uint16_t height)
    : m_output{width, height}, m_blockGrid{m_output}, m_currentBlock{m_blockGrid.begin()}, m_gridWidth{width},
      m_gridHeight{height} {}

// Stores the valid part of the block at the current position (block_grid.cpp:87-99).
void OutputBlockGrid::processNextBlock(BlockGrid::Block const& inputBlock) {
    const size_t offset = static_cast<size_t>(m_currentBlock.getDataPtr() - m_blockGrid.begin().getDataPtr());
    const size_t x0 = offset % m_gridWidth, y0 = offset / m_gridWidth;
    const size_t cols = std::min<size_t>(8, m_gridWidth - x0), rows = std::min<size_t>(8, m_gridHeight - y0);
    for (size_t r = 0; r < rows; ++r)
        std::memcpy(m_output.m_imageData.data() + offset + r * m_gridWidth, inputBlock.m_blockPixelData.data() + 8 * r,
                    cols * sizeof(BitmapImageRGB::PixelData));
    ++m_currentBlock;
}

BitmapImageRGB OutputBlockGrid::getBitmapRGB() const {
    if (m_currentBlock != m_blockGrid.end())
        std::cerr << "Warning: not all JPEG blocks processed. Decoded Bitmap may be incomplete!\n";
    return m_output;
}

bool OutputBlockGrid::atEnd() const { return m_currentBlock == m_blockGrid.end(); }

// ---- colour mapping ----------------------------------------------------------------

ColourMappedBlockData ColourMapper::map(BlockGrid::Block const& inputBlock) const { return applyMapping(inputBlock); }
BlockGrid::Block ColourMapper::unmap(ColourMappedBlockData const& inputBlock) const { return reverseMapping(inputBlock); }
bool ColourMapper::isLuminanceComponent(uint8_t component) const { return componentIsLuminance(component); }

ColourMappedBlockData RGBToRGBMapper::applyMapping(BlockGrid::Block const& in) const {
    ColourMappedBlockData out;
    for (size_t i = 0; i < 64; ++i) {
        out.m_data[0][i] = in.m_blockPixelData[i].r;
        out.m_data[1][i] = in.m_blockPixelData[i].g;
        out.m_data[2][i] = in.m_blockPixelData[i].b;
    }
    return out;
}
BlockGrid::Block RGBToRGBMapper::reverseMapping(ColourMappedBlockData const& in) const {
    BlockGrid::Block out;
    for (size_t i = 0; i < 64; ++i) out.m_blockPixelData[i] = {in.m_data[0][i], in.m_data[1][i], in.m_data[2][i]};
    return out;
}
bool RGBToRGBMapper::componentIsLuminance(uint8_t) const { return true; }

// colour_mapping.cpp:44-67 on the GPU: 64 interleaved pixels in, planar channels out.
ColourMappedBlockData RGBToYCbCrMapper::applyMapping(BlockGrid::Block const& in) const {
    static_assert(sizeof(BitmapImageRGB::PixelData) == 3);
    uint8_t ycc[192];
    {
        std::lock_guard<std::mutex> lock(g_component_mutex);
        jea_ctx* ctx = componentContext();
        check(ctx, jea_rgb_to_ycbcr_host(ctx, reinterpret_cast<uint8_t const*>(in.m_blockPixelData.data()), 64, ycc));
    }
    ColourMappedBlockData out;
    for (size_t i = 0; i < 64; ++i)
        for (size_t c = 0; c < 3; ++c) out.m_data[c][i] = ycc[3 * i + c];
    return out;
}

BlockGrid::Block RGBToYCbCrMapper::reverseMapping(ColourMappedBlockData const& in) const {
    uint8_t ycc[192];
    for (size_t i = 0; i < 64; ++i)
        for (size_t c = 0; c < 3; ++c) ycc[3 * i + c] = in.m_data[c][i];
    BlockGrid::Block out;
    std::lock_guard<std::mutex> lock(g_component_mutex);
    jea_ctx* ctx = componentContext();
    check(ctx, jea_ycbcr_to_rgb_host(ctx, ycc, 64, reinterpret_cast<uint8_t*>(out.m_blockPixelData.data())));
    return out;
}
bool RGBToYCbCrMapper::componentIsLuminance(uint8_t component) const { return component == 0; }

// ---- DCT -----------------------------------------------------------------------------

DctBlockChannelData DiscreteCosineTransformer::transform(ColourMappedBlockData::BlockChannelData const& inputChannel) const {
    return applyTransform(applyOffset(inputChannel));
}
ColourMappedBlockData::BlockChannelData DiscreteCosineTransformer::inverseTransform(DctBlockChannelData const& inputChannel) const {
    return removeOffset(applyInverseTransform(inputChannel));
}
std::array<int8_t, 64> DiscreteCosineTransformer::applyOffset(ColourMappedBlockData::BlockChannelData const& input) const {
    std::array<int8_t, 64> out;
    for (size_t i = 0; i < 64; ++i) out[i] = static_cast<int8_t>(input[i] ^ 0x80);  // u8 - 128 as a bit flip
    return out;
}
ColourMappedBlockData::BlockChannelData DiscreteCosineTransformer::removeOffset(std::array<int8_t, 64> const& input) const {
    ColourMappedBlockData::BlockChannelData out;
    for (size_t i = 0; i < 64; ++i) out[i] = static_cast<uint8_t>(input[i]) ^ 0x80;
    return out;
}

// The device entry points take and return the unshifted u8 samples (they do the
// level shift themselves), so the int8 interface values are flipped back and forth.
DctBlockChannelData SeparatedDiscreteCosineTransformer::applyTransform(std::array<int8_t, 64> const& inputChannel) const {
    uint8_t samples[64];
    for (size_t i = 0; i < 64; ++i) samples[i] = static_cast<uint8_t>(inputChannel[i]) ^ 0x80;
    DctBlockChannelData out;
    std::lock_guard<std::mutex> lock(g_component_mutex);
    jea_ctx* ctx = componentContext();
    check(ctx, jea_dct_forward_blocks(ctx, samples, 1, out.m_data.data()));
    return out;
}
std::array<int8_t, 64> SeparatedDiscreteCosineTransformer::applyInverseTransform(DctBlockChannelData const& inputChannel) const {
    uint8_t samples[64];
    {
        std::lock_guard<std::mutex> lock(g_component_mutex);
        jea_ctx* ctx = componentContext();
        check(ctx, jea_dct_inverse_blocks(ctx, inputChannel.m_data.data(), 1, samples));
    }
    std::array<int8_t, 64> out;
    for (size_t i = 0; i < 64; ++i) out[i] = static_cast<int8_t>(samples[i] ^ 0x80);
    return out;
}

// ---- quantiser -------------------------------------------------------------------------

Quantiser::Quantiser(int quality) {
    jea_quant_tables(quality, m_luminanceQuantisationMatrix.data(), m_chrominanceQuantisationMatrix.data());
}

QuantisedBlockChannelData Quantiser::quantise(DctBlockChannelData const& dctInput, bool useLuminanceMatrix) const {
    QuantisedBlockChannelData out;
    std::lock_guard<std::mutex> lock(g_component_mutex);
    jea_ctx* ctx = componentContext();
    auto const& table = useLuminanceMatrix ? m_luminanceQuantisationMatrix : m_chrominanceQuantisationMatrix;
    check(ctx, jea_quantise_blocks(ctx, dctInput.m_data.data(), 1, table.data(), out.m_data.data()));
    return out;
}

DctBlockChannelData Quantiser::dequantise(QuantisedBlockChannelData const& quantisedInput, bool useLuminanceMatrix) const {
    DctBlockChannelData out;
    std::lock_guard<std::mutex> lock(g_component_mutex);
    jea_ctx* ctx = componentContext();
    auto const& table = useLuminanceMatrix ? m_luminanceQuantisationMatrix : m_chrominanceQuantisationMatrix;
    check(ctx, jea_dequantise_blocks(ctx, quantisedInput.m_data.data(), 1, table.data(), out.m_data.data()));
    return out;
}

// DQT segment: the bytes 20..153 of the library's header (quantiser.cpp:87-102).
void Quantiser::encodeHeaderQuantisationTables(BitStream& outputStream) const {
    uint8_t header[JEA_HEADER_BYTES];
    jea_write_header(header, 8, 8, m_luminanceQuantisationMatrix.data(), m_chrominanceQuantisationMatrix.data());
    for (size_t i = 20; i < 154; ++i) outputStream.pushByte(header[i]);
}

// ---- entropy coding ------------------------------------------------------------------------

HuffmanEncoder::HuffmanEncoder() = default;

// DHT segment: bytes 173..592 of the header (entropy_encoder.cpp:636-694).
void HuffmanEncoder::encodeHeaderEntropyTables(BitStream& outputStream) const {
    uint16_t ones[64];
    for (auto& v : ones) v = 1;
    uint8_t header[JEA_HEADER_BYTES];
    jea_write_header(header, 8, 8, ones, ones);
    for (size_t i = 173; i < 593; ++i) outputStream.pushByte(header[i]);
}

// entropy_encoder.cpp:3-13: zig-zag, run-length form, then the subclass's final encoding -- and back.
void EntropyEncoder::encode(QuantisedBlockChannelData const& input, int16_t& lastDCValue, BitStream& outputStream,
                            bool isLuminanceComponent) const {
    applyFinalEncoding(applyRunLengthEncoding(mapFromGridToZigZag(input), lastDCValue), outputStream, isLuminanceComponent);
}

QuantisedBlockChannelData EntropyEncoder::decode(BitStream const& inputStream, BitStreamReadProgress& readProgress,
                                                 int16_t& lastDCValue, bool isLuminanceComponent) const {
    return mapFromZigZagToGrid(removeRunLengthEncoding(removeFinalEncoding(inputStream, readProgress, isLuminanceComponent),
                                                       lastDCValue));
}

// :15-92 -- a change of layout only
QuantisedBlockChannelData EntropyEncoder::mapFromGridToZigZag(QuantisedBlockChannelData const& input) const {
    QuantisedBlockChannelData out;
    for (size_t k = 0; k < 64; ++k) out.m_data[k] = input.m_data[kZigzag[k]];
    return out;
}
QuantisedBlockChannelData EntropyEncoder::mapFromZigZagToGrid(QuantisedBlockChannelData const& input) const {
    QuantisedBlockChannelData out;
    for (size_t k = 0; k < 64; ++k) out.m_data[kZigzag[k]] = input.m_data[k];
    return out;
}

// :94-120.  An entry is made for every non-zero coefficient and for every sixteenth zero in a row; when the
// block ends in a zero, the entries of trailing zeros go and the end-of-block entry (0, 0) comes instead.
RunLengthEncodedBlockChannelData EntropyEncoder::applyRunLengthEncoding(QuantisedBlockChannelData const& input,
                                                                        int16_t& lastDCValue) const {
    RunLengthEncodedBlockChannelData out;
    out.m_dcDifference = static_cast<int16_t>(input.m_data[0] - lastDCValue);
    lastDCValue = input.m_data[0];
    size_t zeros = 0;
    for (size_t k = 1; k < 64; ++k) {
        const int16_t v = input.m_data[k];
        if (v == 0 && zeros < 15) {
            ++zeros;
            continue;
        }
        out.m_acCoefficients.push_back({zeros, v});
        zeros = 0;
    }
    if (input.m_data[63] == 0) {
        while (!out.m_acCoefficients.empty() && out.m_acCoefficients.back().m_value == 0) out.m_acCoefficients.pop_back();
        out.m_acCoefficients.push_back({0, 0});
    }
    return out;
}

// :122-144.  Every entry but the LAST is restored (:129), so a block that ends without the end-of-block entry
// loses its final coefficient -- kept, the decoded bitmaps depend on it.
QuantisedBlockChannelData EntropyEncoder::removeRunLengthEncoding(RunLengthEncodedBlockChannelData const& input,
                                                                  int16_t& lastDCValue) const {
    QuantisedBlockChannelData out;
    out.m_data.fill(0);
    out.m_data[0] = static_cast<int16_t>(input.m_dcDifference + lastDCValue);
    lastDCValue = out.m_data[0];
    size_t at = 1;
    for (size_t i = 0; i + 1 < input.m_acCoefficients.size(); ++i) {
        at += input.m_acCoefficients[i].m_runLength;
        if (at < 64) out.m_data[at] = input.m_acCoefficients[i].m_value;
        ++at;
    }
    return out;
}

// :696-701, 715-765 on the GPU: Huffman codes and amplitude bits of the entries as they are given.
void HuffmanEncoder::applyFinalEncoding(RunLengthEncodedBlockChannelData const& input, BitStream& outputStream,
                                        bool isLuminanceComponent) const {
    const size_t n = input.m_acCoefficients.size();
    std::vector<uint16_t> runs(n);
    std::vector<int16_t> values(n);
    for (size_t i = 0; i < n; ++i) {
        // the reference narrows the run to 8 bits before it looks at it (entropy_encoder.cpp:733): a run of 256 with
        // value 0 is an end-of-block entry there, and so it is here
        runs[i] = static_cast<uint16_t>(input.m_acCoefficients[i].m_runLength & 0xff);
        values[i] = input.m_acCoefficients[i].m_value;
    }
    std::vector<uint32_t> bits((22 + 26 * n + 31) / 32);
    uint32_t length = 0;
    {
        std::lock_guard<std::mutex> lock(g_component_mutex);
        jea_ctx* ctx = componentContext();
        check(ctx, jea_huffman_encode_rle(ctx, input.m_dcDifference, runs.data(), values.data(), n, isLuminanceComponent ? 1 : 0,
                                          &length, bits.data(), bits.size()));
    }
    for (size_t done = 0; done < length;) {
        const size_t take = std::min<size_t>(16, length - done);
        const size_t word = done / 32, shift = done % 32;
        const uint64_t two = (static_cast<uint64_t>(bits[word]) << 32) | (word + 1 < bits.size() ? bits[word + 1] : 0u);
        outputStream.pushBitsu16(static_cast<uint16_t>((two << shift) >> (64 - take)), take);
        done += take;
    }
}

// :703-712, 767-866 on the GPU.
RunLengthEncodedBlockChannelData HuffmanEncoder::removeFinalEncoding(BitStream const& inputStream,
                                                                     BitStreamReadProgress& readProgress,
                                                                     bool isLuminanceComponent) const {
    // the GPU keeps a copy of the stream; it is refreshed when the contents differ from the uploaded ones
    static bool uploaded = false;
    static uint64_t uploadedRevision = 0;
    uint16_t runs[64];
    int16_t values[64];
    int16_t dcDifference = 0;
    size_t n = 0;
    uint64_t endBit = 0;
    {
        std::lock_guard<std::mutex> lock(g_component_mutex);
        jea_ctx* ctx = componentContext();
        if (!uploaded || uploadedRevision != inputStream.revision()) {
            check(ctx, jea_stream_upload(ctx, inputStream.getDataPtr(), inputStream.committedBytes()));
            uploaded = true;
            uploadedRevision = inputStream.revision();
        }
        check(ctx, jea_huffman_decode_rle(ctx, 8 * static_cast<uint64_t>(readProgress.currentByte) + readProgress.currentBit,
                                          isLuminanceComponent ? 1 : 0, &dcDifference, runs, values, &n, &endBit));
    }
    readProgress.currentByte = static_cast<size_t>(endBit / 8);
    readProgress.currentBit = static_cast<size_t>(endBit % 8);
    RunLengthEncodedBlockChannelData out;
    out.m_dcDifference = dcDifference;
    for (size_t i = 0; i < n; ++i) out.m_acCoefficients.push_back({runs[i], values[i]});
    return out;
}

// ---- per-image Huffman tables -----------------------------------------------------------

OptimisedHuffmanEncoder::OptimisedHuffmanEncoder() = default;

namespace {
[[noreturn]] void perImageTablesNeedTheWholeImage() {
    throw std::runtime_error("OptimisedHuffmanEncoder codes whole images only (its tables need a pass over all blocks first)");
}
}  // namespace

void OptimisedHuffmanEncoder::encodeHeaderEntropyTables(BitStream&) const { perImageTablesNeedTheWholeImage(); }

void OptimisedHuffmanEncoder::applyFinalEncoding(RunLengthEncodedBlockChannelData const&, BitStream&, bool) const {
    perImageTablesNeedTheWholeImage();
}

RunLengthEncodedBlockChannelData OptimisedHuffmanEncoder::removeFinalEncoding(BitStream const&, BitStreamReadProgress&,
                                                                              bool) const {
    perImageTablesNeedTheWholeImage();
}

// ---- encoder -----------------------------------------------------------------------------

Encoder::Encoder(std::unique_ptr<ColourMapper> colourMapper,
                 std::unique_ptr<DiscreteCosineTransformer> discreteCosineTransformer,
                 std::unique_ptr<Quantiser> quantiser, std::unique_ptr<EntropyEncoder> entropyEncoder)
    : m_colourMapper{std::move(colourMapper)}, m_discreteCosineTransformer{std::move(discreteCosineTransformer)},
      m_quantiser{std::move(quantiser)}, m_entropyEncoder{std::move(entropyEncoder)} {}

Encoder::~Encoder() {
    if (m_context) jea_destroy(m_context);
}

void Encoder::setDevice(int deviceId) {
    if (m_context) { jea_destroy(m_context); m_context = nullptr; }
    m_device = deviceId;
}

jea_ctx* Encoder::context() {
    if (!m_context && jea_create(m_device, &m_context) != JEA_OK)
        throw std::runtime_error(std::string("GPU context: ") + jea_last_error(nullptr));
    // the library's colour mode follows the injected mapper
    const bool passThrough = m_colourMapper && m_colourMapper->deviceCapability() == DeviceColourMapping::rgbPassThrough;
    if (jea_set_colour_mapper(m_context, passThrough ? JEA_MAPPER_RGB : JEA_MAPPER_YCBCR) != JEA_OK)
        throw std::runtime_error(std::string("GPU context: ") + jea_last_error(m_context));
    // ... and its Huffman tables follow the injected entropy coder
    const bool perImage = m_entropyEncoder && m_entropyEncoder->deviceCapability() == DeviceEntropyCoding::perImageHuffman;
    if (jea_set_huffman_optimised(m_context, perImage ? 1 : 0) != JEA_OK)
        throw std::runtime_error(std::string("GPU context: ") + jea_last_error(m_context));
    return m_context;
}

void Encoder::requireComponents() const {
    if (!m_colourMapper || !m_discreteCosineTransformer || !m_quantiser || !m_entropyEncoder)
        throw std::runtime_error("Encoder is missing a component");
}

// The fused GPU path implements the library's own components, and only those: a subclass of one of them may
// override any step, so the test is on the exact type as well as on the capability it reports.
bool Encoder::allOnDevice() const {
    auto const& mapper = *m_colourMapper;
    auto const& transformer = *m_discreteCosineTransformer;
    auto const& coder = *m_entropyEncoder;
    const bool knownMapper = (typeid(mapper) == typeid(RGBToYCbCrMapper) && mapper.deviceCapability() == DeviceColourMapping::rgbToYCbCr) ||
                             (typeid(mapper) == typeid(RGBToRGBMapper) && mapper.deviceCapability() == DeviceColourMapping::rgbPassThrough);
    return knownMapper &&
           typeid(transformer) == typeid(SeparatedDiscreteCosineTransformer) && transformer.deviceCapability() == DeviceTransform::separatedFloat &&
           ((typeid(coder) == typeid(HuffmanEncoder) && coder.deviceCapability() == DeviceEntropyCoding::annexKHuffman) ||
            (typeid(coder) == typeid(OptimisedHuffmanEncoder) && coder.deviceCapability() == DeviceEntropyCoding::perImageHuffman));
}

// Encoder::encode as the reference runs it (encoder.cpp:16-37, header :74-138): every step through the injected
// components' per-block methods, so that a user's subclass is honoured.  The library's own components do their
// arithmetic on the GPU, one small launch per call -- a path for correctness, not for speed.
void Encoder::encodePerBlock(BitmapImageRGB const& inputImage, JPEGImage& outputImage) {
    BitStream& stream = outputImage.m_compressedImageData;
    uint16_t ones[64];
    for (auto& v : ones) v = 1;
    uint8_t fixed[JEA_HEADER_BYTES];
    jea_write_header(fixed, inputImage.m_width, inputImage.height, ones, ones);
    for (size_t i = 0; i < 20; ++i) stream.pushByte(fixed[i]);            // SOI, APP0
    m_quantiser->encodeHeaderQuantisationTables(stream);                   // DQT
    for (size_t i = 154; i < 173; ++i) stream.pushByte(fixed[i]);         // SOF0
    m_entropyEncoder->encodeHeaderEntropyTables(stream);                   // DHT, the injected coder's
    for (size_t i = 593; i < JEA_HEADER_BYTES; ++i) stream.pushByte(fixed[i]);   // SOS
    const size_t startOfScanData = stream.getSize();
    InputBlockGrid grid(inputImage);
    std::array<int16_t, 3> lastDC = {0, 0, 0};
    for (auto block = grid.begin(); block != grid.end(); ++block) {
        const ColourMappedBlockData mapped = m_colourMapper->map(*block);
        for (size_t channel = 0; channel < 3; ++channel) {
            const bool lum = m_colourMapper->isLuminanceComponent(static_cast<uint8_t>(channel));
            const DctBlockChannelData dct = m_discreteCosineTransformer->transform(mapped.m_data[channel]);
            m_entropyEncoder->encode(m_quantiser->quantise(dct, lum), lastDC[channel], stream, lum);
        }
    }
    stream.stuffBytes(startOfScanData);
    stream.pushIntoAlignment();
    stream.pushWord(markerEndOfImageSegmentEOI);
}

// Encoder::decode likewise (encoder.cpp:46-66); the header has been validated by the caller.
void Encoder::decodePerBlock(JPEGImage& inputImage, BitmapImageRGB& outputImage) {
    OutputBlockGrid grid(inputImage.m_width, inputImage.m_height);
    BitStreamReadProgress progress;
    progress.currentByte = JEA_HEADER_BYTES;
    BitStream& stream = inputImage.m_compressedImageData;
    stream.removeStuffedBytes(progress);
    std::array<int16_t, 3> lastDC = {0, 0, 0};
    while (!grid.atEnd()) {
        ColourMappedBlockData block;
        for (size_t channel = 0; channel < 3; ++channel) {
            const bool lum = m_colourMapper->isLuminanceComponent(static_cast<uint8_t>(channel));
            const QuantisedBlockChannelData quantised = m_entropyEncoder->decode(stream, progress, lastDC[channel], lum);
            block.m_data[channel] = m_discreteCosineTransformer->inverseTransform(m_quantiser->dequantise(quantised, lum));
        }
        grid.processNextBlock(m_colourMapper->unmap(block));
    }
    if (stream.readNextAlignedWord(progress) != markerEndOfImageSegmentEOI) throw std::runtime_error("Failed to find EOI marker");
    outputImage = grid.getBitmapRGB();
}

// Encoder::encode of the reference (encoder.cpp:14-42): same outputs, same
// "[Error]: ..." convention; the work is one call into the GPU library.
void Encoder::encode(BitmapImageRGB const& inputImage, JPEGImage& outputImage) {
    try {
        outputImage.m_compressedImageData.clearStream();
        requireComponents();
        if (inputImage.m_width == 0 || inputImage.height == 0 ||
            inputImage.m_imageData.size() != static_cast<size_t>(inputImage.m_width) * inputImage.height)
            throw std::runtime_error("Input bitmap is empty or inconsistent");
        if (!allOnDevice()) {
            encodePerBlock(inputImage, outputImage);
            outputImage.m_width = inputImage.m_width;
            outputImage.m_height = inputImage.height;
            outputImage.m_fileSize = outputImage.m_compressedImageData.getSize();
            outputImage.m_supportsSaving = supportsSaving();
            return;
        }
        jea_ctx* ctx = context();
        // worst case is some 19 bytes per pixel: the buffer is left uninitialised, so that only the pages the
        // file really takes are ever touched
        const size_t capacity = jea_max_jpeg_bytes(inputImage.m_width, inputImage.height);
        std::unique_ptr<uint8_t[]> file(new uint8_t[capacity]);
        size_t fileLen = 0;
        check(ctx, jea_encode_host(ctx, reinterpret_cast<uint8_t const*>(inputImage.m_imageData.data()),
                                   inputImage.m_width, inputImage.height, m_quantiser->luminanceTable().data(),
                                   m_quantiser->chrominanceTable().data(), file.get(), capacity, &fileLen));
        outputImage.m_compressedImageData.appendBytes(file.get(), fileLen);
        outputImage.m_width = inputImage.m_width;
        outputImage.m_height = inputImage.height;
        outputImage.m_fileSize = outputImage.m_compressedImageData.getSize();
        outputImage.m_supportsSaving = supportsSaving();
    } catch (std::exception const& e) {
        std::cout << "[Error]: " << e.what() << "\n";
    }
}

// Encoder::decode (encoder.cpp:44-71): dimensions come from the JPEGImage fields,
// tables from this encoder's quantiser; header fields are validated with the
// reference's messages; the output is only replaced on success.
void Encoder::decode(JPEGImage inputImage, BitmapImageRGB& outputImage) {
    try {
        requireComponents();
        BitStream& stream = inputImage.m_compressedImageData;
        const size_t len = stream.committedBytes();
        uint16_t headerWidth = 0, headerHeight = 0;
        const char* message = nullptr;
        if (jea_check_header(stream.getDataPtr(), len, &headerWidth, &headerHeight, &message) != JEA_OK)
            throw std::runtime_error(message);
        outputImage.height = headerHeight;  // encoder.cpp:199-200; overwritten below on success
        outputImage.m_width = headerWidth;
        if (inputImage.m_width == 0 || inputImage.m_height == 0) throw std::runtime_error("JPEG image has no dimensions");
        if (!allOnDevice()) {
            decodePerBlock(inputImage, outputImage);
            return;
        }
        jea_ctx* ctx = context();
        // The library touches the caller's pixels only when the whole decode has succeeded, so an output
        // that already has the right size is decoded into directly; otherwise into a fresh bitmap.
        const size_t pixels = static_cast<size_t>(inputImage.m_width) * inputImage.m_height;
        if (outputImage.m_imageData.size() == pixels) {
            check(ctx, jea_decode_host(ctx, stream.getDataPtr(), len, inputImage.m_width, inputImage.m_height,
                                       m_quantiser->luminanceTable().data(), m_quantiser->chrominanceTable().data(),
                                       reinterpret_cast<uint8_t*>(outputImage.m_imageData.data())));
            outputImage.m_width = inputImage.m_width;
            outputImage.height = inputImage.m_height;
            outputImage.m_fileSize = 0;   // as a bitmap made by the (width, height) constructor
        } else {
            BitmapImageRGB decoded(inputImage.m_width, inputImage.m_height);
            check(ctx, jea_decode_host(ctx, stream.getDataPtr(), len, inputImage.m_width, inputImage.m_height,
                                       m_quantiser->luminanceTable().data(), m_quantiser->chrominanceTable().data(),
                                       reinterpret_cast<uint8_t*>(decoded.m_imageData.data())));
            outputImage = std::move(decoded);
        }
    } catch (std::exception const& e) {
        std::cout << "[Error]: " << e.what() << "\n";
    }
}

}  // namespace jpeg
