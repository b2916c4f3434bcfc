// Host-side anti-aliasing sample tables (no GPU involved).
//
// Follows /root/reference/VisualRayTrace/Assets/Scripts/RayTracer/Sampler/*Sampler.cs and
// AbstractSampler.cs:20-52,143-182.  The table returned is already in access order: entry
// (set p, i) is samples[p*n + shuffled[p*n + i]], which is what the i-th SampleUnitSquare call
// of a pixel that drew set p returns (AbstractSampler.cs:163).  UnityEngine.Random is replaced
// by PCG32 (XSH-RR): Range(int a, int b) = a + next() % (b - a), Range(float a, float b) =
// a + (b - a) * (next() >> 8) * 2^-24.  The base-class constructor (shuffled indices) draws
// before the derived GenerateSamples, as in C#.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/vrt.h"

namespace {

class Pcg32 {
public:
    explicit Pcg32(uint64_t seed) : state_(0), inc_((54u << 1) | 1u)
    {
        next(); state_ += seed; next();
    }
    uint32_t next()
    {
        uint64_t old = state_;
        state_ = old * 6364136223846793005ULL + inc_;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    int range(int a, int b) { return b > a ? a + (int)(next() % (uint32_t)(b - a)) : a; }
    float range(float a, float b) { return a + (b - a) * ((float)(next() >> 8) * (1.f / 16777216.f)); }

private:
    uint64_t state_, inc_;
};

struct Vec2 { float x, y; };

// HammersleySampler.cs:22-35 (a float-valued variant of the radical inverse)
float phi(float j, float vStep)
{
    float x = 0.f, f = vStep * 0.5f;
    while (j >= 0.5f) {
        x += f * std::fmod(j * vStep, 2.f);
        j *= 0.5f;
        f *= 0.5f;
    }
    return x;
}

} // namespace

extern "C" int vrt_sampler_table(int sampler, int n, int sets, float hStep, float vStep, uint64_t seed,
                                 float* out)
{
    if (n <= 0 || sets <= 0 || !out) return VRT_ERR_INVALID;
    if (sampler < VRT_SAMPLER_REGULAR || sampler > VRT_SAMPLER_HAMMERSLEY) return VRT_ERR_INVALID;
    Pcg32 rng(seed);

    // AbstractSampler.SetupShuffledIndices: one Fisher-Yates pass per set over the same list
    std::vector<int> shuffled;
    shuffled.reserve((size_t)n * sets);
    {
        std::vector<int> idx(n);
        for (int j = 0; j < n; ++j) idx[j] = j;
        for (int p = 0; p < sets; ++p) {
            for (int m = n; m > 1;) {
                --m;
                int k = rng.range(0, m + 1);
                std::swap(idx[k], idx[m]);
            }
            shuffled.insert(shuffled.end(), idx.begin(), idx.end());
        }
    }

    std::vector<Vec2> samples;
    samples.reserve((size_t)n * sets);
    const int root = (int)std::sqrt((double)(float)n);
    switch (sampler) {
    case VRT_SAMPLER_REGULAR:
        for (int s = 0; s < sets; ++s)
            for (int p = 0; p < root; ++p)
                for (int q = 0; q < root; ++q)
                    samples.push_back({(q * hStep) / (float)root, (p * vStep) / (float)root});
        break;
    case VRT_SAMPLER_RANDOM:
        for (int i = 0; i < n * sets; ++i) {
            float h = rng.range(0.f, hStep - 1e-5f);
            float v = rng.range(0.f, vStep - 1e-5f);
            samples.push_back({h, v});
        }
        break;
    case VRT_SAMPLER_JITTERED:
        for (int s = 0; s < sets; ++s)
            for (int j = 0; j < root; ++j)
                for (int k = 0; k < root; ++k) {
                    float h = rng.range(0.f, hStep - 1e-5f);
                    float v = rng.range(0.f, vStep - 1e-5f);
                    samples.push_back({(k * hStep + h) / (float)root, (j * vStep + v) / (float)root});
                }
        break;
    case VRT_SAMPLER_NROOKS:
        // the x/y shuffles of NRooksSampler.cs:33-83 write to copies of the list's structs and
        // change nothing, so the diagonal pattern survives
        for (int s = 0; s < sets; ++s)
            for (int j = 0; j < n; ++j) {
                float h = rng.range(0.f, hStep - 1e-5f);
                float v = rng.range(0.f, vStep - 1e-5f);
                samples.push_back({(j * hStep + h) / (float)n, (j * vStep + v) / (float)n});
            }
        break;
    case VRT_SAMPLER_MULTIJITTERED:
        // MultiJitteredSampler.cs:23-27,39: pre-filled with (0,0), every later Set() hits a copy
        samples.assign((size_t)n * sets, Vec2{0.f, 0.f});
        break;
    case VRT_SAMPLER_HAMMERSLEY:
        for (int s = 0; s < sets; ++s)
            for (int j = 0; j < n; ++j)
                samples.push_back({(float)j * hStep / (float)n, phi(j * vStep, vStep)});
        break;
    }

    for (int p = 0; p < sets; ++p)
        for (int i = 0; i < n; ++i) {
            size_t at = (size_t)p * n + i;
            size_t k = (size_t)p * n + (size_t)shuffled[at];
            // past the end the C# indexer throws and SampleUnitSquare returns the zero vector
            Vec2 v = k < samples.size() ? samples[k] : Vec2{0.f, 0.f};
            out[2 * at] = v.x;
            out[2 * at + 1] = v.y;
        }
    return VRT_OK;
}
