// sela_cli.cpp -- command line front end of the MI355X SELA host:
//   sela_mi355x -e in.wav out.sela     encode
//   sela_mi355x -d in.sela out.wav     decode
//   sela_mi355x -E out_dir [--gpus N | --devices a,b,..] a.wav b.wav ...    encode many files as one job -> out_dir/<name>.sela
//   sela_mi355x -D out_dir [--gpus N | --devices a,b,..] a.sela b.sela ...  decode many files as one job -> out_dir/<name>.wav
//   (-E / -D also take --io-threads N: threads that read and write files beside the GPU workers)
//   sela_mi355x -p in.sela [out.pcm]   play: the packets the reference's player would hand to libao (src/sela/player.cpp:30-62),
//                                      handed out while the file is still being decoded -- to out.pcm, or to standard output
//                                      (`sela_mi355x -p in.sela | aplay -f S16_LE -c 2 -r 44100`; this build has no audio device)
// Same verbs as the reference CLI (src/main.cpp:16-27).  The batch verbs spread the files' frames over the GPUs of the
// node (default: all of them), one host thread each.
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "sela_host/codec.hpp"
#include "sela_host/player.hpp"

namespace {

int usage(const std::string& program)
{
    std::cout << "Usage:\n\n"
              << "Encoding a file:\n" << program << " -e path/to/input.wav path/to/output.sela\n\n"
              << "Decoding a file:\n" << program << " -d path/to/input.sela path/to/output.wav\n\n"
              << "Playing a file (raw interleaved int16 to a file, or to standard output):\n" << program << " -p path/to/input.sela [path/to/output.pcm]\n\n"
              << "Many files, all GPUs:\n" << program << " -E|-D path/to/output_dir [--gpus N | --devices 0,1,..] inputs...\n";
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
    std::vector<std::string> inputs;
    for (int i = 3; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--io-threads" && i + 1 < argc) {
            sela::setIoThreads((unsigned)std::max(0, std::atoi(argv[++i])));
        } else if ((a == "--gpus" || a == "--devices") && i + 1 < argc) {
            std::vector<int> devs;
            const std::string v = argv[++i];
            if (a == "--gpus") {
                for (int d = 0; d < std::atoi(v.c_str()); d++)
                    devs.push_back(d);
            } else {
                for (size_t at = 0; at <= v.size();) {
                    const size_t comma = std::min(v.find(',', at), v.size());
                    devs.push_back(std::atoi(v.substr(at, comma - at).c_str()));
                    at = comma + 1;
                }
            }
            if (devs.empty())
                throw data::Exception("no device selected");
            sela::setDevices(devs);
        } else {
            inputs.push_back(a);
        }
    }
    if (inputs.empty())
        return 2;
    // nothing is read or written here: every GPU worker reads, codes and writes its own pieces of the files
    std::vector<std::string> outputs;
    for (const std::string& in : inputs)
        outputs.push_back(sibling(out_dir, in, verb == "-E" ? ".sela" : ".wav"));
    if (verb == "-E")
        sela::encodeFiles(inputs, outputs);
    else
        sela::decodeFiles(inputs, outputs);
    return 0;
}

int run(int argc, char** argv)
{
    const std::string program = argv[0];
    const std::string verb = argc > 1 ? argv[1] : "";
    if ((verb == "-E" || verb == "-D") && argc >= 4) {
        const int rc = batch(verb, argc, argv);
        return rc == 2 ? usage(program) : rc;
    }
    if (verb == "-p" && (argc == 3 || argc == 4)) {
        int fd = STDOUT_FILENO;
        if (argc == 4) {
            fd = ::open(argv[3], O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
            if (fd < 0)
                throw data::Exception(std::string("cannot open ") + argv[3] + " for writing");
        } else if (::isatty(STDOUT_FILENO)) {
            std::cerr << "-p writes raw samples to standard output: redirect it, or name an output file" << std::endl;
            return 2;
        }
        std::cerr << "Playing: " << argv[2] << std::endl;
        sela::RawPcmSink sink(fd);
        sela::Player player(sink);
        player.showProgress = ::isatty(STDERR_FILENO) != 0;
        size_t frames = 0;
        try {
            frames = player.playFile(argv[2]);
        } catch (...) {
            if (fd != STDOUT_FILENO)
                (void)::close(fd);
            throw;
        }
        if (fd != STDOUT_FILENO)
            (void)::close(fd);
        std::cerr << frames << " frames, first packet after " << player.firstPacketSeconds * 1e3 << " ms" << std::endl;
        return 0;
    }
    if (argc != 4 || (verb != "-e" && verb != "-d"))
        return usage(program);
    if (verb == "-e") {
        std::cout << "Encoding: " << argv[2] << std::endl;
        sela::encodeFile(std::string(argv[2]), std::string(argv[3])); // read, GPU and write overlap; only the byte stream is produced
    } else {
        std::cout << "Decoding: " << argv[2] << std::endl;
        sela::decodeFile(std::string(argv[2]), std::string(argv[3]));
    }
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    // (-p may write the samples to standard output: everything else it says goes to standard error)
    (argc > 1 && std::string(argv[1]) == "-p" ? std::cerr : std::cout) << "SimplE Lossless Audio (.sela v2 bitstream) -- MI355X host" << std::endl;
    try {
        return run(argc, argv);
    } catch (const data::Exception& e) {
        std::cerr << e.exceptionMessage << std::endl;
    } catch (const std::exception& e) { // (std::bad_alloc on an absurd header, ...)
        std::cerr << "error: " << e.what() << std::endl;
    }
    return 1;
}
