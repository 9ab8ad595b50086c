// sela_cli.cpp -- command line front end of the MI355X SELA host:
//   sela_mi355x -e in.wav out.sela     encode
//   sela_mi355x -d in.sela out.wav     decode
// Same verbs as the reference CLI (src/main.cpp:16-27); playback (-p) is not part of this build.
#include <fstream>
#include <iostream>
#include <string>

#include "sela_host/codec.hpp"

namespace {

int usage(const std::string& program)
{
    std::cout << "Usage:\n\n"
              << "Encoding a file:\n" << program << " -e path/to/input.wav path/to/output.sela\n\n"
              << "Decoding a file:\n" << program << " -d path/to/input.sela path/to/output.wav\n";
    return 2;
}

} // namespace

int main(int argc, char** argv)
{
    std::cout << "SimplE Lossless Audio (.sela v2 bitstream) -- MI355X host" << std::endl;
    const std::string program = argv[0];
    if (argc != 4)
        return usage(program);
    const std::string verb = argv[1];
    try {
        std::ifstream in(argv[2], std::ios::binary);
        if (!in)
            throw data::Exception(std::string("cannot open ") + argv[2]);
        if (verb == "-e") {
            std::cout << "Encoding: " << argv[2] << std::endl;
            sela::Encoder::materializeFrames = false; // only the byte stream is needed
            file::SelaFile sela = sela::Encoder(in).process();
            std::ofstream out(argv[3], std::ios::binary);
            sela.writeToFile(out);
        } else if (verb == "-d") {
            std::cout << "Decoding: " << argv[2] << std::endl;
            sela::Decoder::demuxFrames = false;
            file::WavFile wav = sela::Decoder(in).process();
            std::ofstream out(argv[3], std::ios::binary);
            wav.writeToFile(out);
        } else {
            return usage(program);
        }
    } catch (const data::Exception& e) {
        std::cerr << e.exceptionMessage << std::endl;
        return 1;
    }
    return 0;
}
