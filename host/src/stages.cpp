// stages.cpp -- see stages.hpp.
#include "sela_host/stages.hpp"

#include <algorithm>
#include <string>

#include "sela_format.h"
#include "sela_hip.h"

namespace {

constexpr size_t kBlock = SELA_HIP_SAMPLES_PER_FRAME;
constexpr size_t kMaxOrder = 100;

[[noreturn]] void stageFailure(const char* what)
{
    throw data::Exception(std::string(what) + ": " + sela_hip_last_error());
}

} // namespace

namespace rice {

data::RiceEncodedData RiceEncoder::process()
{
    const uint64_t valueOffsets[2] = { 0, input.size() };
    uint64_t wordOffsets[2] = { 0, 0 };
    uint32_t k = 0, words = 0;
    // the word count first (no room: the call reports SELA_HIP_ECAPACITY and the count), then the words
    int rc = sela_hip_rice_encode(input.data(), valueOffsets, 1, &k, &words, nullptr, wordOffsets);
    if (rc != SELA_HIP_OK && rc != SELA_HIP_ECAPACITY)
        stageFailure("RiceEncoder");
    std::vector<uint32_t> out(words);
    wordOffsets[1] = words;
    if (words && sela_hip_rice_encode(input.data(), valueOffsets, 1, &k, &words, out.data(), wordOffsets) != SELA_HIP_OK)
        stageFailure("RiceEncoder");
    return data::RiceEncodedData(k, (uint32_t)input.size(), std::move(out));
}

data::RiceDecodedData RiceDecoder::process()
{
    const uint64_t wordOffsets[2] = { 0, input.size() }, valueOffsets[2] = { 0, dataCount };
    std::vector<int32_t> out(dataCount);
    if (sela_hip_rice_decode(input.data(), wordOffsets, &optimumRiceParam, valueOffsets, 1, out.data()) != SELA_HIP_OK)
        stageFailure("RiceDecoder");
    return data::RiceDecodedData(std::move(out));
}

} // namespace rice

namespace lpc {

void LinearPredictor::generatelinearPredictionCoefficients()
{
    const int32_t order = optimalLpcOrder;
    if ((size_t)order > kMaxOrder || quantizedReflectionCoefficients.size() < (size_t)order)
        throw data::Exception("LinearPredictor: order beyond 100 or fewer coefficients than the order");
    std::vector<int32_t> q(kMaxOrder, 0);
    for (int32_t i = 0; i < order; i++)
        q[(size_t)i] = quantizedReflectionCoefficients[(size_t)i];
    std::vector<int64_t> a(kMaxOrder + 1, 0);
    if (sela_hip_lpc_decode(&order, q.data(), nullptr, 1, nullptr, a.data()) != SELA_HIP_OK)
        stageFailure("LinearPredictor");
    linearPredictionCoefficients.assign(a.begin(), a.begin() + order + 1);
}

// src/lpc/linear_predictor.cpp:16-28: order <= 1 -> [0.0]; else the three tables (include/sela_tables.inc, the reference's
// data), indexed with q + 64.  (Host arithmetic there is none: three lookups per coefficient.  An index outside the tables --
// undefined in the reference -- is clamped like the kernels clamp it.)
void LinearPredictor::dequantizeReflectionCoefficients()
{
    reflectionCoefficients.clear();
    if (optimalLpcOrder <= 1) {
        reflectionCoefficients.push_back(0.0);
        return;
    }
    if (quantizedReflectionCoefficients.size() < (size_t)optimalLpcOrder)
        throw data::Exception("LinearPredictor: fewer coefficients than the order");
    auto at = [](int32_t q) { return (size_t)std::min(127, std::max(0, (int)std::min<int64_t>(std::max<int64_t>((int64_t)q + 64, -1), 128))); };
    reflectionCoefficients.reserve(optimalLpcOrder);
    reflectionCoefficients.push_back(SELA_DEQUANT_FIRST[at(quantizedReflectionCoefficients[0])]);
    reflectionCoefficients.push_back(SELA_DEQUANT_SECOND[at(quantizedReflectionCoefficients[1])]);
    for (size_t i = 2; i < (size_t)optimalLpcOrder; i++)
        reflectionCoefficients.push_back(SELA_DEQUANT_HIGHER[at(quantizedReflectionCoefficients[i])]);
}

// (any number of samples of any 32-bit value, like the reference's class: 2048 samples go to the frame kernels' analysis,
// other lengths to the any-length kernels; a block not longer than the order its analysis picks is refused -- the reference
// reads past its vector there, src/lpc/residue_generator.cpp:104-110)
data::LpcEncodedData ResidueGenerator::process()
{
    const size_t n = samples.size();
    if (n == 0 || n > ((size_t)1 << 24))
        throw data::Exception("ResidueGenerator: 1 .. 2^24 samples");
    int32_t order = 0;
    std::vector<int32_t> q(kMaxOrder, 0), residues(n);
    const int rc = n == kBlock ? sela_hip_lpc_encode(samples.data(), 1, &order, q.data(), residues.data())
                               : sela_hip_lpc_encode_n(samples.data(), 1, (uint32_t)n, &order, q.data(), residues.data());
    if (rc != SELA_HIP_OK)
        stageFailure("ResidueGenerator");
    q.resize((size_t)order);
    return data::LpcEncodedData((uint8_t)order, bitsPerSample, std::move(q), std::move(residues));
}

data::LpcDecodedData SampleGenerator::process()
{
    const size_t n = residues.size();
    if (n == 0 || n > ((size_t)1 << 24))
        throw data::Exception("SampleGenerator: 1 .. 2^24 residues");
    const int32_t order = linearPredictor.optimalLpcOrder;
    if ((size_t)order > kMaxOrder || linearPredictor.quantizedReflectionCoefficients.size() < (size_t)order)
        throw data::Exception("SampleGenerator: order beyond 100 or fewer coefficients than the order");
    std::vector<int32_t> q(kMaxOrder, 0), out(n);
    for (int32_t i = 0; i < order; i++)
        q[(size_t)i] = linearPredictor.quantizedReflectionCoefficients[(size_t)i];
    const int rc = n == kBlock ? sela_hip_lpc_decode(&order, q.data(), residues.data(), 1, out.data(), nullptr)
                               : sela_hip_lpc_decode_n(&order, q.data(), residues.data(), 1, (uint32_t)n, out.data(), nullptr);
    if (rc != SELA_HIP_OK)
        stageFailure("SampleGenerator");
    return data::LpcDecodedData(bitsPerSample, std::move(out));
}

} // namespace lpc
