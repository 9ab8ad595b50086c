// stages.hpp -- the reference's L1 classes (src/include/lpc.hpp:73-117, src/include/rice.hpp:9-43) over the stage entries of
// libsela_hip.so (sela_hip_lpc_encode / _decode, sela_hip_rice_encode / _decode): the same constructors and process()
// signatures, one block or stream per object, computed on the MI355X.
//
// These are NOT the fast path: a frame goes through all four stages inside one kernel (frame::FrameEncoder /
// FrameDecoder, sela::Encoder / Decoder), and each call here is a device round trip of its own.  They exist for callers
// and tests of a stage by itself -- the reference's own tests (test/lpctests.cpp:10-32, test/ricetests.cpp:7-25) are
// written against these classes.  No CPU fallback: without a device process() throws data::Exception.
#pragma once

#include <cstdint>
#include <vector>

#include "sela_host/data.hpp"

constexpr uint8_t MAX_RICE_PARAM = 20; // src/include/rice.hpp:7

namespace rice {

class RiceEncoder {
    const std::vector<int32_t>& input;

public:
    explicit RiceEncoder(const data::RiceDecodedData& decodedData) : input(decodedData.decodedData) {}
    // src/rice/rice_encoder.cpp:73-81.  LIMITS: |value| < 2^30 (the reference's int32 zig-zag overflows beyond: data::Exception);
    // a stream of more than 2^24 bits comes back with the word count the reference's float arithmetic gives it (:37,63), its
    // last bits unwritten like the reference's.
    data::RiceEncodedData process();
};

class RiceDecoder {
    const std::vector<uint32_t>& input;
    uint32_t dataCount, optimumRiceParam;

public:
    explicit RiceDecoder(const data::RiceEncodedData& encodedData)
        : input(encodedData.encodedData), dataCount(encodedData.dataCount), optimumRiceParam(encodedData.optimumRiceParam)
    {
    }
    data::RiceDecodedData process(); // src/rice/rice_decoder.cpp:54-61
};

} // namespace rice

namespace lpc {

// src/include/lpc.hpp:73-85.  dequantizeReflectionCoefficients() fills reflectionCoefficients from the reference's tables (three
// lookups per coefficient; the device looks the same tables up for itself); generatelinearPredictionCoefficients() fills the
// Q35 predictor on the device.
class LinearPredictor {
public:
    uint8_t optimalLpcOrder = 1;
    std::vector<double> reflectionCoefficients;
    std::vector<int64_t> linearPredictionCoefficients;
    std::vector<int32_t> quantizedReflectionCoefficients;
    LinearPredictor() {}
    LinearPredictor(std::vector<int32_t> quantizedReflectionCoefficients, uint8_t optimalLpcOrder)
        : optimalLpcOrder(optimalLpcOrder), quantizedReflectionCoefficients(std::move(quantizedReflectionCoefficients))
    {
    }
    void dequantizeReflectionCoefficients();      // src/lpc/linear_predictor.cpp:16-28
    void generatelinearPredictionCoefficients(); // src/lpc/linear_predictor.cpp:30-61
};

class ResidueGenerator {
    const std::vector<int32_t>& samples;
    uint8_t bitsPerSample;

public:
    explicit ResidueGenerator(const data::LpcDecodedData& data) : samples(data.samples), bitsPerSample(data.bitsPerSample) {}
    // src/lpc/residue_generator.cpp:121-134.  Any number of samples (1 .. 2^24) of any 32-bit value.  LIMITS: a block must be
    // longer than the order its own analysis picks -- the reference reads past its vector otherwise (:104-110) -- else
    // data::Exception; a residue Rice stream too long for a frame's slot is not this stage's business (the residues are
    // returned whatever their size).
    data::LpcEncodedData process();
};

class SampleGenerator {
    const std::vector<int32_t>& residues;
    uint8_t bitsPerSample;
    LinearPredictor linearPredictor;

public:
    explicit SampleGenerator(const data::LpcEncodedData& encodedData)
        : residues(encodedData.residues), bitsPerSample(encodedData.bitsPerSample),
          linearPredictor(encodedData.quantizedReflectionCoefficients, encodedData.optimalLpcOrder)
    {
    }
    data::LpcDecodedData process(); // src/lpc/sample_generator.cpp:32-39
};

} // namespace lpc
