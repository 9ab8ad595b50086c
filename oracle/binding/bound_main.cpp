// oracle/binding/bound_main.cpp -- TEST INFRASTRUCTURE: the two file verbs of the reference's command line over the
// reference's own classes (what src/main.cpp:29-41 does with them), for the binary `make -C oracle bound` links:
//     sela_ref_bound -e in.wav out.sela        sela_ref_bound -d in.sela out.wav
#include <fstream>
#include <iostream>
#include <string>

#include "data/exception.hpp"
#include "sela/decoder.hpp"
#include "sela/encoder.hpp"

int main(int argc, char** argv)
{
    if (argc != 4 || (std::string(argv[1]) != "-e" && std::string(argv[1]) != "-d")) {
        std::cerr << "usage: " << argv[0] << " -e in.wav out.sela | -d in.sela out.wav" << std::endl;
        return 2;
    }
    try {
        std::ifstream in(argv[2], std::ios::binary);
        if (!in) {
            std::cerr << "cannot open " << argv[2] << std::endl;
            return 1;
        }
        std::ofstream out(argv[3], std::ios::binary);
        if (std::string(argv[1]) == "-e") {
            sela::Encoder encoder(in);
            file::SelaFile selaFile = encoder.process();
            selaFile.writeToFile(out);
        } else {
            sela::Decoder decoder(in);
            file::WavFile wavFile = decoder.process();
            wavFile.writeToFile(out);
        }
    } catch (const data::Exception& e) {
        std::cerr << e.exceptionMessage << std::endl;
        return 1;
    }
    return 0;
}
