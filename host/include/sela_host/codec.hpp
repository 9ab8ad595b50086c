// codec.hpp -- sela::Encoder / sela::Decoder with the reference's signatures
// (src/include/sela/encoder.hpp:9-22, src/include/sela/decoder.hpp:9-22).
//
// Where the reference reads the whole file and then fans the frames out to hardware_concurrency() threads
// (src/sela/encoder.cpp:40-100), these stream the file through the MI355X: the file is read piece by
// piece into page-locked memory and every piece is handed to the GPU (sela_hip_encode_feed /
// sela_hip_decode_feed) while the next one is still being read.  There is no CPU fallback: if the GPU path
// is unavailable process() throws data::Exception.
#pragma once

#include <fstream>
#include <string>
#include <vector>

#include "sela_host/files.hpp"

namespace sela {

class Encoder {
    std::ifstream& ifStream;
    file::WavFile wavFile;

public:
    // Also build the data::SelaFrame objects in SelaFile::selaFrames (API fidelity).  Tools that only
    // write the file can switch this off: the byte stream is complete without them.
    static bool materializeFrames;
    explicit Encoder(std::ifstream& in) : ifStream(in) {}
    file::SelaFile process();
};

class Decoder {
    std::ifstream& ifStream;
    file::SelaFile selaFile;

public:
    static bool demuxFrames; // also fill WavFile::wavFrames (per-frame de-interleaved copies)
    explicit Decoder(std::ifstream& in) : ifStream(in) {}
    file::WavFile process();
};

// File to file (what the reference's main.cpp:29-41 does with process() + writeToFile()): the same
// streaming read, and finished frames / samples are written out while later pieces are still on the device.
// Return the number of frames coded.
size_t encodeFile(std::ifstream& in, std::ofstream& out);
size_t decodeFile(std::ifstream& in, std::ofstream& out);
// The same by path -- what the CLI's -e / -d use: the file is read with several pread()s in flight on a small pool of I/O
// threads (sela_host/fileio.hpp) while earlier pieces are on the device, and finished ranges are written by a task of
// that pool -- over pages allocated in one go while the input was still on its way -- while later pieces are being coded.
// One thread reads or writes a page-cache file at a few GB/s; the device codes 10 G samples/s.
size_t encodeFile(const std::string& inPath, const std::string& outPath);
size_t decodeFile(const std::string& inPath, const std::string& outPath);
// Decoding for a consumer that takes the samples in order (the player, sela_host/player.hpp): begin() once the header is
// known, then ready(pcm, n) -- on the calling thread -- whenever the first n interleaved samples of pcm are final (n only
// grows; pcm is the caller's buffer `into`, sized here before the first call, so the samples outlive the decoding: a player is
// still handing them out long after the last frame was decoded).  Returns the frames decoded.
class DecodedStream {
public:
    virtual ~DecodedStream() {}
    virtual void begin(const data::SelaHeader& header, size_t announcedFrames) = 0;
    virtual void ready(const int16_t* pcm, size_t samples) = 0;
};
size_t decodeFileTo(const std::string& inPath, DecodedStream& to, sela_host::PinnedBuffer<int16_t>& into);
// Threads of that pool (before its first use; 0 = default: min(6, hardware threads / 2)).
void setIoThreads(unsigned n);

// ---- many files, all GPUs of the node ------------------------------------------------------------------
// BASELINE.json configs[3]: an album's tracks are one index space of frames, cut into contiguous balanced
// ranges -- the reference's static partition (src/sela/encoder.cpp:58-73) with the remainder spread -- one
// per device, one host thread per device.  A range may begin or end inside a track.  The compressed sizes
// of the pieces meet in host memory (this is one process; sela_amd/sharding.py is the one-process-per-GPU
// form with the RCCL all-gather).  Results are what Encoder / Decoder give file by file.
//
// Devices to use: setDevices({0, 1, ...}); an index may repeat (two workers on one GPU -- used by the
// tests).  Default: every visible device.
void setDevices(const std::vector<int>& devices);
std::vector<int> devices();
std::vector<file::SelaFile> encodeBatch(const std::vector<file::WavFile>& wavs);
std::vector<file::WavFile> decodeBatch(const std::vector<file::SelaFile>& selas);
// The same jobs by path (CLI -E / -D): nothing is read or written up front -- every GPU worker reads ITS pieces of the
// input files (read-ahead on the I/O pool), codes them, and writes the results at their place in the output files
// while its next piece is on the device.  Small tracks that follow each other are read into one buffer and coded as
// one job.  A track cut between two workers: the later piece is placed when the earlier one's size is known; no
// worker waits for another one before its own work is done.
void encodeFiles(const std::vector<std::string>& inputs, const std::vector<std::string>& outputs);
void decodeFiles(const std::vector<std::string>& inputs, const std::vector<std::string>& outputs);

} // namespace sela
