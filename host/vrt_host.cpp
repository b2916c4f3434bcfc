// Implementation of the C++ host layer (see vrt_host.hpp).  Everything that computes runs in
// libvrt.so; this file only marshals.
#include "vrt_host.hpp"

#include <cmath>
#include <cstring>

namespace vrt {

// ---------------------------------------------------------------- AntiAliasingStrategy

AntiAliasingStrategy::AntiAliasingStrategy(AASamplingStrategy strat, int sampleSize, int numSets, float hStep,
                                           float vStep, uint64_t seed)
    : sampleSize_(sampleSize > 0 ? sampleSize : 1), numSets_(numSets > 0 ? numSets : 1) // AntiAliasingStrategy.cs:34
{
    offsets_.assign((size_t)sampleSize_ * numSets_ * 2, 0.f);
    int rc = vrt_sampler_table((int)strat, sampleSize_, numSets_, hStep, vStep, seed, offsets_.data());
    if (rc != VRT_OK) throw Error(rc, "vrt_sampler_table failed");
}

std::vector<Vector3> AntiAliasingStrategy::CreateAARays(const Vector3& initRay, uint32_t pixel, uint32_t setSeed) const
{
    // AntiAliasingStrategy.cs:83-95; the sample set of a pixel is vrt_pixel_set(seed, pixel, sets)
    std::vector<Vector3> rays;
    rays.reserve(sampleSize_);
    uint32_t set = vrt_pixel_set(setSeed, pixel, (uint32_t)numSets_);
    for (int i = 0; i < sampleSize_; ++i) {
        const float* o = &offsets_[2 * ((size_t)set * sampleSize_ + i)];
        Vector3 r; r.x = initRay.x + o[0]; r.y = initRay.y + o[1]; r.z = initRay.z;
        rays.push_back(r);
    }
    return rays;
}

// ---------------------------------------------------------------- WorldInformation

WorldInformation::WorldInformation() { vrt_shading_defaults(&shading); }

int WorldInformation::AddMaterial(const std::string& name, const Color& color)
{
    // Util/RayTraceUtility.cs:139-152: the first word of the Unity material name decides
    std::string first = name.substr(0, name.find(' '));
    vrt_material m;
    m.type = first == "Metal" ? VRT_MAT_METAL : first == "Dielectric" ? VRT_MAT_DIELECTRIC : VRT_MAT_SOLID;
    m.color[0] = color.r; m.color[1] = color.g; m.color[2] = color.b; m.color[3] = color.a;
    materials_.push_back(m);
    return (int)materials_.size() - 1;
}

void WorldInformation::AddSphere(const Vector3& c, float radius, int material)
{
    vrt_sphere s; s.center[0] = c.x; s.center[1] = c.y; s.center[2] = c.z; s.radius = radius; s.material = material;
    spheres_.push_back(s);
}

void WorldInformation::AddBox(const Vector3& c, const Vector3 axes[3], const Vector3& half, int material)
{
    vrt_box b;
    b.center[0] = c.x; b.center[1] = c.y; b.center[2] = c.z;
    for (int k = 0; k < 3; ++k) { b.axis[3 * k] = axes[k].x; b.axis[3 * k + 1] = axes[k].y; b.axis[3 * k + 2] = axes[k].z; }
    b.half[0] = half.x; b.half[1] = half.y; b.half[2] = half.z; b.material = material;
    boxes_.push_back(b);
}

void WorldInformation::AddDirectionalLight(const Vector3& dir, float intensity)
{
    // colour forced white, shadows off, _location never assigned (Light/DirectionalLight.cs:15-20)
    vrt_light l; memset(&l, 0, sizeof l);
    l.type = VRT_LIGHT_DIRECTIONAL; l.vec[0] = dir.x; l.vec[1] = dir.y; l.vec[2] = dir.z; l.ls = intensity;
    l.color[0] = l.color[1] = l.color[2] = l.color[3] = 1.f; l.cast_shadows = 0;
    lights_.push_back(l);
}

void WorldInformation::AddPointLight(float intensity, const Color& color, const Vector3& p)
{
    vrt_light l; memset(&l, 0, sizeof l);
    l.type = VRT_LIGHT_POINT; l.vec[0] = p.x; l.vec[1] = p.y; l.vec[2] = p.z; l.ls = intensity;
    l.color[0] = color.r; l.color[1] = color.g; l.color[2] = color.b; l.color[3] = color.a; l.cast_shadows = 1;
    lights_.push_back(l);
}

vrt_scene_desc WorldInformation::Desc() const
{
    vrt_scene_desc d; memset(&d, 0, sizeof d);
    d.n_spheres = (uint32_t)spheres_.size(); d.spheres = spheres_.data();
    d.n_boxes = (uint32_t)boxes_.size(); d.boxes = boxes_.data();
    d.n_materials = (uint32_t)materials_.size(); d.materials = materials_.data();
    d.n_lights = (uint32_t)lights_.size(); d.lights = lights_.data();
    d.shading = shading;
    return d;
}

// ---------------------------------------------------------------- Device

Device::Device(int ordinal)
{
    int rc = vrt_create(ordinal, &ctx_);
    if (rc != VRT_OK) throw Error(rc, "vrt_create failed: no usable HIP device");
}
Device::~Device() { if (ctx_) vrt_destroy(ctx_); }
void Device::Check(int rc) const { if (rc != VRT_OK) throw Error(rc, vrt_last_error(ctx_)); }

// ---------------------------------------------------------------- WhittedTracer

WhittedTracer::WhittedTracer(Device& dev, const WorldInformation& world, int /*maxDepth: ignored like the C#*/,
                             float maxDist)
    : dev_(dev), world_(world)
{
    vrt_params_defaults(&params_);
    params_.range = maxDist;
    vrt_scene_desc d = world.Desc();
    dev_.Check(vrt_set_scene(dev_.ctx(), &d));
}

Color WhittedTracer::TraceRay(const Ray& ray, int depth)
{
    params_.max_depth = world_.MaxDepth;
    float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    float c[4]; int32_t id = -1;
    dev_.Check(vrt_trace_ray(dev_.ctx(), &params_, o, d, depth, c, &id));
    lastHit_ = id;
    Color out; out.r = c[0]; out.g = c[1]; out.b = c[2]; out.a = c[3];
    return out;
}

// ---------------------------------------------------------------- RayTracerUnity

RayTracerUnity::RayTracerUnity(Device& dev, const WorldInformation& world) : dev_(dev), world_(world)
{
    vrt_params_defaults(&params_);
    memset(&stats_, 0, sizeof stats_);
}

RayTracerUnity::~RayTracerUnity() { delete aa_; }

void RayTracerUnity::Start()
{
    // InitPlane / InitTexture / SetIterationSteps (:273-277, Info/ViewPortPlaneInformation.cs:110-117):
    // the steps are the plane extents over the pixel counts; the C# takes the extents from the
    // renderer bounds (size.z and size.y, :362-363), which for its viewport are the axis lengths
    auto len = [](const Vector3& v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); };
    float hStep = (float)(len(planeXAxis) / textureWidth), vStep = (float)(len(planeYAxis) / textureHeight);
    delete aa_;
    aa_ = new AntiAliasingStrategy(SamplingMethod, SampleSize, SampleSetCount, hStep, vStep, 1); // :293
    vrt_scene_desc d = world_.Desc();                                                            // :299
    dev_.Check(vrt_set_scene(dev_.ctx(), &d));
    vrt_camera cam;
    cam.origin[0] = rayOrigin.x; cam.origin[1] = rayOrigin.y; cam.origin[2] = rayOrigin.z;
    cam.p0[0] = planeP0.x; cam.p0[1] = planeP0.y; cam.p0[2] = planeP0.z;
    cam.x_axis[0] = planeXAxis.x; cam.x_axis[1] = planeXAxis.y; cam.x_axis[2] = planeXAxis.z;
    cam.y_axis[0] = planeYAxis.x; cam.y_axis[1] = planeYAxis.y; cam.y_axis[2] = planeYAxis.z;
    cam.width = textureWidth; cam.height = textureHeight;
    dev_.Check(vrt_set_camera(dev_.ctx(), &cam));
    params_.spp = aa_->SampleSize(); params_.n_sets = aa_->NumSets(); params_.offsets = aa_->Offsets().data();
    params_.seed = 1; params_.max_depth = world_.MaxDepth; params_.range = RayTrace_Range; // :300-306
    pixels_.assign((size_t)textureWidth * textureHeight, Color{0, 0, 0, 1}); // Info/TextureInformation.cs:58-63
    hitIds_.assign(pixels_.size(), -1);
}

void RayTracerUnity::RenderAll()
{
    dev_.Check(vrt_render(dev_.ctx(), &params_, &pixels_[0].r, WantHitIds ? hitIds_.data() : nullptr, &stats_));
}

void RayTracerUnity::SetTexturePixel(int x, int y, const Color& c)
{
    if (x < 0 || x > textureHeight - 1) return; // :438-439 (x is the vertical coordinate there)
    if (y < 0 || y > textureWidth - 1) return;
    pixels_[(size_t)x * textureWidth + y] = c;  // :442
}

Color RayTracerUnity::DoRayTraceArchitecture(int hCord, int vCord)
{
    const int cap = 253; // the C# counts polyline points in a byte starting at 2 (:649)
    segs_.assign(VisualizePath ? cap : 0, vrt_path_seg{});
    int n = 0; float c[4]; int32_t id = -1;
    dev_.Check(vrt_trace_pixel(dev_.ctx(), &params_, hCord, vCord, -1, VisualizePath ? segs_.data() : nullptr,
                               VisualizePath ? cap : 0, &n, c, &id));
    segs_.resize(VisualizePath ? (size_t)(n < cap ? n : cap) : 0);
    Color out; out.r = c[0]; out.g = c[1]; out.b = c[2]; out.a = c[3];
    SetTexturePixel(hCord, vCord, out); // UpdateScene, :607-617
    if (hCord >= 0 && hCord < textureHeight && vCord >= 0 && vCord < textureWidth)
        hitIds_[(size_t)hCord * textureWidth + vCord] = id;
    return out;
}

std::vector<Vector3> RayTracerUnity::RT_rec_points() const
{
    // origin, first hit, then every further hit point of the sample-0 reflection/transmission
    // chain: what the LineRenderer of UpdateScene shows (:620-654)
    std::vector<Vector3> pts;
    for (const vrt_path_seg& s : segs_) {
        if (s.sample != 0) break;
        if (s.kind != VRT_RAY_PRIMARY && s.kind != VRT_RAY_REFLECT && s.kind != VRT_RAY_TRANSMIT) continue;
        if (pts.empty()) pts.push_back(Vector3{s.origin[0], s.origin[1], s.origin[2]});
        pts.push_back(Vector3{s.end[0], s.end[1], s.end[2]});
    }
    return pts;
}

} // namespace vrt
