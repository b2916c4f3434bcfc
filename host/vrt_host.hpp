// C++ host layer over the C ABI of libvrt.so, shaped like the reference's tracer-facing
// classes so that a port of its C# host code (or a test written against it) reads the same.
// The reference is C# on Unity; no C# toolchain exists in this build environment, so this
// layer stands where `VrtNative.cs` + the patched RayTracerUnity would stand.
//
// Mirrors (paths under VisualRayTrace/Assets/Scripts/RayTracer/):
//   Vector3 / Color / Ray            UnityEngine value types as the tracer uses them
//   AASamplingStrategy, AntiAliasingStrategy   AntiAliasingStrategy.cs:8,13-96
//   AbstractTracer, WhittedTracer    Tracer/AbstractTracer.cs:5-12, Tracer/WhittedTracer.cs:5-110
//   WorldInformation                 Util/RayTraceUtility.cs:446-530 (scene -> tracer marshalling)
//   RayTracerUnity                   RayTracerUnity.cs: Start / DoRayTraceArchitecture /
//                                    SetTexturePixel / PixelColorData / ray polyline
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/vrt.h"

namespace vrt {

struct Vector3 { float x = 0, y = 0, z = 0; };
struct Color { float r = 0, g = 0, b = 0, a = 1; };
struct Ray { Vector3 origin, direction; }; // direction is normalised by the library, like new Ray()

enum class AASamplingStrategy { Regular, Random, Jittered, NRooks, MultiJittered, Hammersley };

class Error : public std::runtime_error {
public:
    Error(int code, const std::string& what) : std::runtime_error(what), code(code) {}
    int code;
};

// AntiAliasingStrategy.cs:13-96.  Holds the sampler table in access order; CreateAARays adds
// the i-th offset of the pixel's sample set to the x and y of the un-normalised direction.
class AntiAliasingStrategy {
public:
    AntiAliasingStrategy(AASamplingStrategy strat, int sampleSize, int numSets, float hStep, float vStep,
                         uint64_t seed = 0);
    std::vector<Vector3> CreateAARays(const Vector3& initRay, uint32_t pixel, uint32_t setSeed) const;
    int SampleSize() const { return sampleSize_; }
    int NumSets() const { return numSets_; }
    const std::vector<float>& Offsets() const { return offsets_; }

private:
    int sampleSize_, numSets_;
    std::vector<float> offsets_; // numSets * sampleSize * 2
};

// Util/RayTraceUtility.cs:446-530: the scene as the tracer sees it.  Add* calls stand in for the
// collider / material / light discovery the C# does through Unity.
class WorldInformation {
public:
    WorldInformation();
    int MaxDepth = 10;          // :461
    float RaycastDistance = 30; // Util/RayTraceUtility.cs:81
    int AddMaterial(const std::string& unityMaterialName, const Color& color); // name -> class, :139-152
    void AddSphere(const Vector3& center, float radius, int material);
    void AddBox(const Vector3& center, const Vector3 axes[3], const Vector3& half, int material);
    void AddDirectionalLight(const Vector3& dirVector, float intensity);                  // :497-521
    void AddPointLight(float intensity, const Color& color, const Vector3& position);     // :523-525
    vrt_scene_desc Desc() const;
    vrt_shading shading;

private:
    std::vector<vrt_sphere> spheres_;
    std::vector<vrt_box> boxes_;
    std::vector<vrt_material> materials_;
    std::vector<vrt_light> lights_;
};

class Device {
public:
    explicit Device(int ordinal = 0);
    ~Device();
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    vrt_ctx* ctx() const { return ctx_; }
    void Check(int rc) const;

private:
    vrt_ctx* ctx_ = nullptr;
};

// Tracer/AbstractTracer.cs:5-12
class AbstractTracer {
public:
    virtual ~AbstractTracer() = default;
    virtual Color TraceRay(const Ray& ray) = 0;
    virtual Color TraceRay(const Ray& ray, int depth) = 0;
    virtual Color TraceRay(const Ray& ray, float tmin, int depth) = 0;
};

// Tracer/WhittedTracer.cs:5-110 on the GPU.  Like the C#, the constructor's maxDepth argument
// is ignored in favour of the world's MaxDepth (:21-26,37).
class WhittedTracer : public AbstractTracer {
public:
    WhittedTracer(Device& dev, const WorldInformation& world, int maxDepth, float maxDist);
    Color TraceRay(const Ray& ray) override { return TraceRay(ray, world_.MaxDepth); }          // :28-31
    Color TraceRay(const Ray& ray, int depth) override;                                        // :33-104
    Color TraceRay(const Ray& ray, float, int depth) override { return TraceRay(ray, depth); } // :106-109
    int LastHit() const { return lastHit_; }

private:
    Device& dev_;
    const WorldInformation& world_;
    vrt_params params_;
    int lastHit_ = -1;
};

// RayTracerUnity.cs, the parts on the tracing path.
class RayTracerUnity {
public:
    RayTracerUnity(Device& dev, const WorldInformation& world);
    // inspector fields (:88-233)
    float RayTrace_Range = 30.f;
    int SampleSize = 50;
    AASamplingStrategy SamplingMethod = AASamplingStrategy::Hammersley;
    int SampleSetCount = 83;
    bool VisualizePath = true;
    // not in the reference: the primary hit ids of sample 0 are 4 more bytes per pixel over PCIe and
    // nothing on the C# side consumes them, so RenderAll only fetches them when asked to
    bool WantHitIds = false;
    // camera (:285, Info/ViewPortPlaneInformation.cs:86-99)
    Vector3 rayOrigin, planeP0, planeXAxis, planeYAxis;
    int textureWidth = 200, textureHeight = 200; // TextureDimension x TextureDimension in the C#

    void Start();                                // :266-307
    void RenderAll();                            // the Update loop over every pixel, :312-352
    Color DoRayTraceArchitecture(int hCord, int vCord); // one pixel, :565-597; fills RT_rec_points
    const std::vector<Color>& PixelColorData() const { return pixels_; } // Info/TextureInformation.cs:36
    const std::vector<int32_t>& PrimaryHitIds() const { return hitIds_; }
    const std::vector<vrt_path_seg>& RT_rec_segments() const { return segs_; } // Util/RayTraceUtility.cs:154-155
    std::vector<Vector3> RT_rec_points() const; // polyline UpdateScene draws, :620-654
    const vrt_stats& Stats() const { return stats_; }

private:
    void SetTexturePixel(int x, int y, const Color& c); // :435-444
    Device& dev_;
    const WorldInformation& world_;
    AntiAliasingStrategy* aa_ = nullptr;
    vrt_params params_;
    std::vector<Color> pixels_;
    std::vector<int32_t> hitIds_;
    std::vector<vrt_path_seg> segs_;
    vrt_stats stats_;
public:
    ~RayTracerUnity();
};

} // namespace vrt
