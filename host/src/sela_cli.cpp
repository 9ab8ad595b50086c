// sela_cli.cpp -- command line front end of the MI355X SELA host:
//   sela_mi355x -e in.wav out.sela     encode
//   sela_mi355x -d in.sela out.wav     decode
//   sela_mi355x -E out_dir a.wav b.wav ...    encode many files as one GPU batch -> out_dir/<name>.sela
//   sela_mi355x -D out_dir a.sela b.sela ...  decode many files as one GPU batch -> out_dir/<name>.wav
// Same verbs as the reference CLI (src/main.cpp:16-27); playback (-p) is not part of this build.
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "sela_host/codec.hpp"

namespace {

int usage(const std::string& program)
{
    std::cout << "Usage:\n\n"
              << "Encoding a file:\n" << program << " -e path/to/input.wav path/to/output.sela\n\n"
              << "Decoding a file:\n" << program << " -d path/to/input.sela path/to/output.wav\n\n"
              << "Many files in one GPU batch:\n" << program << " -E|-D path/to/output_dir inputs...\n";
    return 2;
}

// out_dir/<file name of `in` with its extension replaced>
std::string sibling(const std::string& out_dir, const std::string& in, const char* extension)
{
    const size_t slash = in.find_last_of('/');
    std::string name = slash == std::string::npos ? in : in.substr(slash + 1);
    const size_t dot = name.find_last_of('.');
    if (dot != std::string::npos)
        name.erase(dot);
    return out_dir + "/" + name + extension;
}

int batch(const std::string& verb, int argc, char** argv)
{
    const std::string out_dir = argv[2];
    if (verb == "-E") {
        std::vector<file::WavFile> wavs((size_t)argc - 3);
        for (int i = 3; i < argc; i++) {
            std::ifstream in(argv[i], std::ios::binary);
            if (!in)
                throw data::Exception(std::string("cannot open ") + argv[i]);
            wavs[(size_t)i - 3].readFromFile(in);
        }
        sela::Encoder::materializeFrames = false;
        std::vector<file::SelaFile> selas = sela::encodeBatch(wavs);
        for (int i = 3; i < argc; i++) {
            std::ofstream out(sibling(out_dir, argv[i], ".sela"), std::ios::binary);
            selas[(size_t)i - 3].writeToFile(out);
        }
    } else {
        std::vector<file::SelaFile> selas((size_t)argc - 3);
        for (int i = 3; i < argc; i++) {
            std::ifstream in(argv[i], std::ios::binary);
            if (!in)
                throw data::Exception(std::string("cannot open ") + argv[i]);
            selas[(size_t)i - 3].readFromFile(in);
        }
        sela::Decoder::demuxFrames = false;
        std::vector<file::WavFile> wavs = sela::decodeBatch(selas);
        for (int i = 3; i < argc; i++) {
            std::ofstream out(sibling(out_dir, argv[i], ".wav"), std::ios::binary);
            wavs[(size_t)i - 3].writeToFile(out);
        }
    }
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    std::cout << "SimplE Lossless Audio (.sela v2 bitstream) -- MI355X host" << std::endl;
    const std::string program = argv[0];
    const std::string verb = argc > 1 ? argv[1] : "";
    if ((verb == "-E" || verb == "-D") && argc >= 4) {
        try {
            return batch(verb, argc, argv);
        } catch (const data::Exception& e) {
            std::cerr << e.exceptionMessage << std::endl;
            return 1;
        }
    }
    if (argc != 4)
        return usage(program);
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
