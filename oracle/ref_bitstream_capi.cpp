// C entry points around the reference's own jpeg::BitStream
// (jpeg/inc/bit_stream.hpp, jpeg/src/bit_stream.cpp), compiled from the
// reference tree as it lies on disk.  TEST INFRASTRUCTURE ONLY: used to check the
// oracle's bit packing, byte stuffing and unstuffing against the real class.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "bit_stream.hpp"

extern "C" {

void* refbs_new() { return new jpeg::BitStream(); }
void refbs_delete(void* p) { delete static_cast<jpeg::BitStream*>(p); }
void refbs_push_bits(void* p, uint16_t data, size_t nbits) { static_cast<jpeg::BitStream*>(p)->pushBitsu16(data, nbits); }
void refbs_push_into_alignment(void* p) { static_cast<jpeg::BitStream*>(p)->pushIntoAlignment(); }
size_t refbs_size(void* p) { return static_cast<jpeg::BitStream*>(p)->getSize(); }
// Copies getSize() bytes: the committed bytes followed by the pending partial byte, if any.
void refbs_copy(void* p, uint8_t* out) {
    auto* s = static_cast<jpeg::BitStream*>(p);
    size_t n = s->getSize();
    for (size_t i = 0; i < n; ++i) out[i] = s->readByte(i);
}
// stuffBytes() keeps using an iterator across vector::insert; it is only safe
// while no insert reallocates, so the caller must size its test streams to leave
// spare capacity (see tests/test_oracle_ref_bitstream.py).
void refbs_stuff(void* p, size_t from) { static_cast<jpeg::BitStream*>(p)->stuffBytes(from); }
void refbs_unstuff(void* p, size_t from) {
    jpeg::BitStreamReadProgress progress;
    progress.currentByte = from;
    static_cast<jpeg::BitStream*>(p)->removeStuffedBytes(progress);
}
int refbs_read_bit(void* p, size_t* byte, size_t* bit) {
    jpeg::BitStreamReadProgress progress;
    progress.currentByte = *byte;
    progress.currentBit = *bit;
    int b = static_cast<jpeg::BitStream*>(p)->readNextBit(progress);
    *byte = progress.currentByte;
    *bit = progress.currentBit;
    return b;
}
}
