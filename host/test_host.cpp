// Self-checking driver for the C++ host layer.
//   test_host --cpu   host logic only (sampler tables, material mapping, error path without a GPU)
//   test_host         needs a GPU: renders the teaching scene through the mirrored classes and
//                     checks the per-pixel and per-ray entry points against the whole frame
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vrt_host.hpp"

using namespace vrt;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static void build_teaching_world(WorldInformation& w)
{
    int metal = w.AddMaterial("Metal (Instance)", Color{0.9528f, 0.9528f, 0.9528f, 1});
    int blue = w.AddMaterial("Blue", Color{0, 0.6747f, 1, 1});
    int red = w.AddMaterial("Red", Color{1, 0, 0, 1});
    int glass = w.AddMaterial("Dielectric (Instance)", Color{1, 1, 1, 1});
    int green = w.AddMaterial("Green", Color{0.4563f, 1, 0, 1});
    int lwood = w.AddMaterial("LightBrownWood", Color{0.8208f, 0.6273f, 0.2671f, 1});
    int white = w.AddMaterial("WhiteEye", Color{1, 1, 1, 1});
    w.AddSphere(Vector3{6.5f, 1.75f, 3.75f}, 0.75f, metal);
    w.AddSphere(Vector3{6.5f, 1.75f, 1.75f}, 0.75f, metal);
    w.AddSphere(Vector3{5.5f, 1.25f, 4.0f}, 0.25f, blue);
    w.AddSphere(Vector3{5.5f, 1.25f, 3.5f}, 0.25f, metal);
    w.AddSphere(Vector3{5.5f, 1.25f, 3.0f}, 0.25f, red);
    w.AddSphere(Vector3{5.5f, 1.25f, 2.5f}, 0.25f, glass);
    w.AddSphere(Vector3{5.5f, 1.25f, 2.0f}, 0.25f, green);
    Vector3 rot[3] = {Vector3{0, 0, -1}, Vector3{0, 1, 0}, Vector3{1, 0, 0}};
    Vector3 id[3] = {Vector3{1, 0, 0}, Vector3{0, 1, 0}, Vector3{0, 0, 1}};
    w.AddBox(Vector3{6.081f, 0.5f, 2.268f}, rot, Vector3{2.75f, 0.5f, 1.25f}, lwood);
    w.AddBox(Vector3{7.417f, 2.0f, 2.219f}, id, Vector3{0.083f, 2.0f, 2.82f}, white);
    w.AddDirectionalLight(Vector3{0, -1, 0}, 1.f);
    w.AddDirectionalLight(Vector3{0.585f, 0.696f, 0.417f}, 1.f);
    w.AddPointLight(1.f, Color{1, 1, 1, 1}, Vector3{4.5f, 5.3f, 2.5f});
}

static void cpu_checks()
{
    // Hammersley as shipped: 50 samples, 83 sets, steps 0.01 / 0.005 -> y == 0, x = j*0.01/50
    AntiAliasingStrategy aa(AASamplingStrategy::Hammersley, 50, 83, 0.01f, 0.005f, 1);
    CHECK(aa.Offsets().size() == 50u * 83u * 2u);
    double sx = 0; bool y0 = true;
    for (int i = 0; i < 50; ++i) { sx += aa.Offsets()[2 * i]; y0 = y0 && aa.Offsets()[2 * i + 1] == 0.f; }
    CHECK(y0);
    CHECK(std::fabs(sx - 0.01 / 50 * (49 * 50 / 2)) < 1e-6);
    std::vector<Vector3> rays = aa.CreateAARays(Vector3{1, 2, 3}, 1234, 7);
    CHECK(rays.size() == 50u);
    CHECK(rays[0].z == 3.f && rays[0].y == 2.f && rays[0].x >= 1.f && rays[0].x <= 1.01f);
    AntiAliasingStrategy one(AASamplingStrategy::Regular, 0, 0, 0.01f, 0.01f); // sampleSize <= 0 -> 1
    CHECK(one.SampleSize() == 1 && one.NumSets() == 1);
    WorldInformation w;
    build_teaching_world(w);
    vrt_scene_desc d = w.Desc();
    CHECK(d.n_spheres == 7 && d.n_boxes == 2 && d.n_lights == 3 && d.n_materials == 7);
    CHECK(d.materials[0].type == VRT_MAT_METAL && d.materials[3].type == VRT_MAT_DIELECTRIC && d.materials[1].type == VRT_MAT_SOLID);
    CHECK(d.lights[0].cast_shadows == 0 && d.lights[2].cast_shadows == 1);
    CHECK(w.shading.metal_kr == 0.75f && w.shading.glass_exp == 1.f);
}

static void gpu_checks()
{
    Device dev(0);
    WorldInformation w;
    build_teaching_world(w);
    w.MaxDepth = 5;
    RayTracerUnity rt(dev, w);
    rt.rayOrigin = Vector3{1.45f, 2, 2.5f};
    rt.planeP0 = Vector3{3, 1.55f, 3.5f};
    rt.planeXAxis = Vector3{0, 0, -2};
    rt.planeYAxis = Vector3{0, 1, 0};
    rt.textureWidth = 96; rt.textureHeight = 64;
    rt.SampleSize = 4; rt.SampleSetCount = 5;
    rt.WantHitIds = true;
    rt.Start();
    rt.RenderAll();
    std::vector<Color> frame = rt.PixelColorData();
    std::vector<int32_t> ids = rt.PrimaryHitIds();
    CHECK(rt.Stats().rays[VRT_RAY_PRECHECK] == 96u * 64u * 4u);
    int hits = 0;
    for (int32_t id : ids) hits += id >= 0;
    CHECK(hits > 96 * 64 / 2);
    // one pixel at a time must reproduce the frame, and record the ray tree
    const int probe[5][2] = {{32, 48}, {10, 10}, {40, 70}, {63, 95}, {20, 30}};
    for (auto& p : probe) {
        Color c = rt.DoRayTraceArchitecture(p[0], p[1]);
        const Color& f = frame[(size_t)p[0] * 96 + p[1]];
        CHECK(std::memcmp(&c, &f, sizeof c) == 0);
        if (ids[(size_t)p[0] * 96 + p[1]] >= 0) {
            CHECK(!rt.RT_rec_segments().empty());
            std::vector<Vector3> pts = rt.RT_rec_points();
            CHECK(pts.size() >= 2);
            CHECK(pts[0].x == 1.45f && pts[0].y == 2.f && pts[0].z == 2.5f);
        }
    }

    // AbstractTracer interface: a ray through the tracer slot
    WhittedTracer tracer(dev, w, w.MaxDepth, 30.f);
    AbstractTracer& t = tracer;
    Ray r; r.origin = rt.rayOrigin; r.direction = Vector3{1.55f, 0.0f, 0.0f};
    Color c0 = t.TraceRay(r, 0);
    CHECK(tracer.LastHit() >= 0);
    CHECK(c0.a == 1.f && c0.r >= 0.f);
    Color deep = t.TraceRay(r, w.MaxDepth + 1); // beyond MaxDepth: black, nothing cast
    CHECK(deep.r == 0.f && deep.g == 0.f && deep.b == 0.f && tracer.LastHit() == -1);
    Ray up; up.origin = rt.rayOrigin; up.direction = Vector3{0, 1, 0};
    Color sky = t.TraceRay(up, 0, 0);
    CHECK(std::fabs(sky.r - 0.5f) < 1e-6f && std::fabs(sky.g - 0.7f) < 1e-6f && sky.b == 1.f);
    // error path: bad pixel
    bool threw = false;
    try { rt.DoRayTraceArchitecture(64, 0); } catch (const Error& e) { threw = e.code == VRT_ERR_INVALID; }
    CHECK(threw);
}

// test_host --dump DIR: what the mirrored classes deliver for the teaching scene, written out so that
// a test can hold it against the CPU oracle: the frame of RenderAll (PixelColorData, primary hit
// ids, ray counts), and for a few pixels what DoRayTraceArchitecture returns, the recorded ray
// tree and the polyline RT_rec_points.  Also the sampler table and the scene as handed to the library.
template <class T>
static void dump(const std::string& dir, const char* name, const T* data, size_t n)
{
    FILE* f = std::fopen((dir + "/" + name).c_str(), "wb");
    if (!f) { std::printf("FAIL: cannot write %s/%s\n", dir.c_str(), name); ++failures; return; }
    if (n) std::fwrite(data, sizeof(T), n, f);
    std::fclose(f);
}

static void dump_run(const std::string& dir)
{
    Device dev(0);
    WorldInformation w;
    build_teaching_world(w);
    w.MaxDepth = 5;
    RayTracerUnity rt(dev, w);
    rt.rayOrigin = Vector3{1.45f, 2, 2.5f};
    rt.planeP0 = Vector3{3, 1.55f, 3.5f};
    rt.planeXAxis = Vector3{0, 0, -2};
    rt.planeYAxis = Vector3{0, 1, 0};
    rt.textureWidth = 96; rt.textureHeight = 64;
    rt.SampleSize = 4; rt.SampleSetCount = 5;
    rt.SamplingMethod = AASamplingStrategy::Jittered;
    rt.WantHitIds = true;
    rt.Start();
    rt.RenderAll();
    vrt_scene_desc d = w.Desc();
    dump(dir, "spheres.bin", d.spheres, d.n_spheres);
    dump(dir, "boxes.bin", d.boxes, d.n_boxes);
    dump(dir, "materials.bin", d.materials, d.n_materials);
    dump(dir, "lights.bin", d.lights, d.n_lights);
    dump(dir, "shading.bin", &d.shading, 1);
    dump(dir, "frame.bin", &rt.PixelColorData()[0].r, rt.PixelColorData().size() * 4);
    dump(dir, "ids.bin", rt.PrimaryHitIds().data(), rt.PrimaryHitIds().size());
    dump(dir, "rays.bin", rt.Stats().rays, (size_t)VRT_RAY_CLASSES);
    const int probe[6][2] = {{32, 48}, {10, 10}, {40, 70}, {63, 95}, {20, 30}, {35, 40}};
    std::vector<int32_t> meta;          // per probe: h, v, hit id, segments, polyline points
    std::vector<float> colors, points;
    std::vector<vrt_path_seg> segs;
    for (auto& p : probe) {
        Color c = rt.DoRayTraceArchitecture(p[0], p[1]);
        std::vector<Vector3> pts = rt.RT_rec_points();
        meta.push_back(p[0]); meta.push_back(p[1]);
        meta.push_back(rt.PrimaryHitIds()[(size_t)p[0] * 96 + p[1]]);
        meta.push_back((int32_t)rt.RT_rec_segments().size()); meta.push_back((int32_t)pts.size());
        colors.push_back(c.r); colors.push_back(c.g); colors.push_back(c.b); colors.push_back(c.a);
        segs.insert(segs.end(), rt.RT_rec_segments().begin(), rt.RT_rec_segments().end());
        for (const Vector3& q : pts) { points.push_back(q.x); points.push_back(q.y); points.push_back(q.z); }
    }
    dump(dir, "probe_meta.bin", meta.data(), meta.size());
    dump(dir, "probe_colors.bin", colors.data(), colors.size());
    dump(dir, "probe_segs.bin", segs.data(), segs.size());
    dump(dir, "probe_points.bin", points.data(), points.size());
    // the tracer slot: WhittedTracer.TraceRay(ray, depth) for a few rays
    WhittedTracer tracer(dev, w, w.MaxDepth, 30.f);
    const float dirs[4][3] = {{1.55f, 0.f, 0.f}, {1.55f, -0.45f, 0.3f}, {0.f, 1.f, 0.f}, {1.55f, -0.2f, -0.6f}};
    std::vector<float> ray_out;
    for (auto& q : dirs)
        for (int depth = 0; depth < 3; ++depth) {
            Ray r; r.origin = rt.rayOrigin; r.direction = Vector3{q[0], q[1], q[2]};
            Color c = tracer.TraceRay(r, depth);
            ray_out.push_back(c.r); ray_out.push_back(c.g); ray_out.push_back(c.b); ray_out.push_back((float)tracer.LastHit());
        }
    dump(dir, "rays_traced.bin", ray_out.data(), ray_out.size());
}

int main(int argc, char** argv)
{
    if (argc > 2 && std::strcmp(argv[1], "--dump") == 0) {
        try { dump_run(argv[2]); } catch (const Error& e) { std::printf("FAIL: libvrt error %d: %s\n", e.code, e.what()); ++failures; }
        std::printf(failures ? "test_host: %d failure(s)\n" : "test_host: OK\n", failures);
        return failures ? 1 : 0;
    }
    bool cpu_only = argc > 1 && std::strcmp(argv[1], "--cpu") == 0;
    cpu_checks();
    if (cpu_only) {
        try { Device d(0); } catch (const Error& e) { CHECK(e.code == VRT_ERR_NO_DEVICE || e.code == VRT_ERR_HIP); }
    } else {
        try { gpu_checks(); } catch (const Error& e) { std::printf("FAIL: libvrt error %d: %s\n", e.code, e.what()); ++failures; }
    }
    std::printf(failures ? "test_host: %d failure(s)\n" : "test_host: OK\n", failures);
    return failures ? 1 : 0;
}
