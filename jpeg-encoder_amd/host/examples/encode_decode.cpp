// Headless counterpart of the reference's demo (examples/encode-decode.cpp): same
// flags (-i input BMP, -o output path, -q quality, -s save the JPEG), same
// encode-then-decode flow and timing printout, no window.  With -s the decoded
// bitmap is also written next to the JPEG as <output>.decoded.bmp.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "encoder.hpp"

int main(int argc, char** argv) {
    std::string input = "img/input.bmp", output = "img/output.jpg";
    int quality = 80;  // the reference's default
    bool save = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "-i") && i + 1 < argc) input = argv[++i];
        else if (!std::strcmp(argv[i], "-o") && i + 1 < argc) output = argv[++i];
        else if (!std::strcmp(argv[i], "-q") && i + 1 < argc) quality = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "-s")) save = true;
        else { std::cerr << "usage: " << argv[0] << " [-i in.bmp] [-o out.jpg] [-q 1..100] [-s]\n"; return 2; }
    }
    jpeg::BitmapImageRGB inputBmp(input);
    if (inputBmp.m_width == 0 || inputBmp.height == 0) return 1;
    std::cout << "Input: " << input << " (" << inputBmp.m_width << "x" << inputBmp.height << ", " << inputBmp.m_fileSize << " bytes)\n";

    jpeg::BaselineEncoder encoder(quality);
    jpeg::JPEGImage outputJpeg;
    auto t0 = std::chrono::high_resolution_clock::now();
    encoder.encode(inputBmp, outputJpeg);
    auto t1 = std::chrono::high_resolution_clock::now();
    jpeg::BitmapImageRGB decodedBmp;
    encoder.decode(outputJpeg, decodedBmp);
    auto t2 = std::chrono::high_resolution_clock::now();
    const double encMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
    const double decMs = std::chrono::duration<double, std::milli>(t2 - t1).count();
    std::cout << "Quality " << quality << ": " << outputJpeg.m_fileSize << " bytes ("
              << (inputBmp.m_fileSize ? 100.0 * double(outputJpeg.m_fileSize) / double(inputBmp.m_fileSize) : 0.0)
              << "% of the input), encode " << encMs << " ms, decode " << decMs << " ms\n";
    if (save) {
        outputJpeg.saveToFile(output);
        decodedBmp.saveToFile(output + ".decoded.bmp");
        std::cout << "Saved " << output << " and " << output << ".decoded.bmp\n";
    }
    return 0;
}
