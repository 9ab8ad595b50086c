// player.hpp -- sela::Player (reference: src/include/sela/player.hpp:9-32, src/sela/player.cpp:30-104), the FEED of it.
//
// The reference decodes the whole file, then a transformer thread turns every WavFrame into one freshly allocated
// interleaved int16 packet (data::AudioPacket, at most 100 ahead of the audio loop) and the audio loop hands the packets
// to libao's ao_play() one by one.  Here the decode kernel already leaves interleaved int16 in page-locked memory, so a
// packet is a VIEW of one frame of it (nothing is transformed or allocated), and playFile() starts handing packets to
// the device while later frames are still being read from the file and decoded: what matters to a player is the time
// to the FIRST packet, not the rate.
//
// The audio device itself is behind sela::AudioSink -- this build has no libao (and the GPU box no sound card); the
// sink a maintainer would add for the reference's behaviour is ten lines around ao_open_live / ao_play / ao_close
// (INTEGRATION.md section 5).  RawPcmSink writes the packets to a file descriptor (pipe it into `aplay -f S16_LE`).
#pragma once

#include <cstddef>
#include <string>

#include "sela_host/files.hpp"

namespace sela {

class AudioSink {
public:
    virtual ~AudioSink() {}
    virtual void open(const data::WavFormatSubChunk& format) = 0; // what setAoFormat() gets (src/sela/player.cpp:20-28)
    virtual void play(const data::AudioPacket& packet) = 0;       // ao_play(dev, packet.audio, packet.bufferSize)
    virtual void close() {}
};

// The packets' bytes, as they are, to a file descriptor (not closed by this class).
class RawPcmSink : public AudioSink {
    int fd;

public:
    explicit RawPcmSink(int fd) : fd(fd) {}
    void open(const data::WavFormatSubChunk&) override {}
    void play(const data::AudioPacket& packet) override; // throws data::Exception when the descriptor cannot take them
};

class Player {
    AudioSink& sink;
    void printProgress(size_t current, size_t total, bool last) const;

public:
    bool showProgress = false; // the reference's progress bar (src/sela/player.cpp:106-131), on std::cerr here
    size_t packetsPlayed = 0;  // of the last play() / playFile()
    double firstPacketSeconds = 0; // playFile(): from the call to the first packet handed to the sink
    explicit Player(AudioSink& s) : sink(s) {}
    // The reference's `sela::Player player;` (src/main.cpp:49): no libao here, so the packets go to standard output as they are
    // (RawPcmSink on descriptor 1: `sela -p in.sela | aplay -f S16_LE ...`).
    Player();
    // One packet per whole frame of wavFile (2048 samples of every channel, interleaved), in order.
    void play(const file::WavFile& wavFile);
    // main.cpp:43-51 (decode, then play) as one overlapped job: returns the number of frames played.
    size_t playFile(const std::string& selaPath);
};

} // namespace sela
