// Frame driver that replaces the per-pixel coroutine of RayTracerUnity
// (RayTracerUnity.cs:312-352,565-597,607-666) by libvrt: the whole frame is traced on the GPU
// without blocking Unity's own frame loop, the single-step mode traces one pixel and feeds the
// ray polyline.  Attach next to RayTracerUnity and hand it the same inspector values; the VR UI
// keeps talking to RayTracerUnity (play / pause / reset), which forwards to StartFrame / Step.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using UnityEngine;

public sealed class NativeRayTraceDriver : MonoBehaviour
{
    public float RayTraceRange = 30f;                                   // RayTrace_Range
    public int SampleSize = 50;                                          // SCN inspector values
    public AASamplingStrategy SamplingMethod = AASamplingStrategy.Hammersley;
    public int SampleSetCount = 83;
    public uint Seed = 1;
    public int Gpus = 1;
    public bool WantHitIds = false;   // primary hit ids of sample 0: 4 more bytes per pixel over PCIe, only fetched when asked for
    public bool Rgba8 = false;        // frames as Color32 (VRT_OUT_RGBA8): what the RGBA32 texture holds after SetPixels anyway (:472-475)
    public Color32[] Pixels32;        // with Rgba8: the frame, for Texture2D.SetPixels32 / LoadRawTextureData

    IntPtr _ctx = IntPtr.Zero;
    VrtNative.Params _prm;
    float[] _offsets; GCHandle _offsetsPin, _pixelsPin, _idsPin;
    int[] _hitIds;
    bool _frameInFlight;
    VrtSceneMarshaller _scene;
    public NativeTracer Tracer { get; private set; }

    /// Camera and buffers come from the objects RayTracerUnity.Start builds (:266-307).
    public void Init(Vector3 rayOrigin, ViewPortPlaneInformation viewPort, TextureInformation texture)
    {
        if (VrtNative.vrt_abi_version() != VrtNative.AbiVersion) throw new InvalidOperationException("libvrt: ABI version mismatch");
        // Gpus > 1: the same calls on a context that spans GPUs 0 .. Gpus-1 of the node (frames in strip sets
        // are a one-GPU call: StartFrameInParts falls back to StartFrame there)
        if (Gpus > 1) VrtNative.Check(IntPtr.Zero, VrtNative.vrt_create_multi(Gpus, null, out _ctx), "vrt_create_multi");
        else VrtNative.Check(IntPtr.Zero, VrtNative.vrt_create(0, out _ctx), "vrt_create");
        RebuildScene();
        int dim = texture.TextureDimension;
        var cam = new VrtNative.Camera { origin = rayOrigin, p0 = viewPort.PlaneBorderPoints[0], xAxis = viewPort.PlaneXAxis,
                                         yAxis = viewPort.PlaneYAxis, width = dim, height = dim };
        VrtNative.Check(_ctx, VrtNative.vrt_set_camera(_ctx, ref cam), "vrt_set_camera");
        // AntiAliasingStrategy constructor (AntiAliasingStrategy.cs:32-67)
        _offsets = new float[SampleSetCount * SampleSize * 2];
        VrtNative.Check(_ctx, VrtNative.vrt_sampler_table((int)SamplingMethod, SampleSize, SampleSetCount,
                                                        viewPort.HorizontalIterationStep, viewPort.VerticalIterationStep, Seed, _offsets), "vrt_sampler_table");
        _offsetsPin = GCHandle.Alloc(_offsets, GCHandleType.Pinned);
        VrtNative.vrt_params_defaults(out _prm);
        _prm.spp = SampleSize; _prm.nSets = SampleSetCount; _prm.offsets = _offsetsPin.AddrOfPinnedObject(); _prm.seed = Seed;
        _prm.maxDepth = RayTraceUtility.GlobalWorld.MaxDepth; _prm.range = RayTraceRange;
        if (Rgba8) { Pixels32 = new Color32[dim * dim]; _prm.flags |= VrtNative.OutRgba8; }
        _pixelsPin = GCHandle.Alloc(Rgba8 ? (object)Pixels32 : texture.PixelColorData, GCHandleType.Pinned);
        if (WantHitIds) { _hitIds = new int[dim * dim]; _idsPin = GCHandle.Alloc(_hitIds, GCHandleType.Pinned); }
        Tracer = new NativeTracer(_ctx, _prm);
        RayTraceUtility.GlobalWorld.Tracer = Tracer;                      // replaces :300-304
    }

    /// After scene edits (spawned spheres, material changes) and ResetRaytracer (:450-456).
    public void RebuildScene()
    {
        _scene = new VrtSceneMarshaller();
        _scene.Collect(RayTraceUtility.LayerMask);
        _scene.Upload(_ctx);
    }

    IntPtr IdsPtr() { return WantHitIds ? _idsPin.AddrOfPinnedObject() : IntPtr.Zero; }

    /// Continuous mode: starts the frame; Update polls it.
    public void StartFrame()
    {
        if (_frameInFlight) return;
        VrtNative.Check(_ctx, VrtNative.vrt_render_begin(_ctx, ref _prm, _pixelsPin.AddrOfPinnedObject(), IdsPtr()), "vrt_render_begin");
        _frameInFlight = true;
    }

    /// Continuous mode with the frame delivered in strip sets: PollParts says how many have landed in
    /// PixelColorData, and the caller uploads only their rows (Texture2D.SetPixels(x, y, w, h, ...) per
    /// strip) instead of SetPixels + Apply of the whole texture (:461-477).
    public int PartsInFlight { get; private set; }
    public void StartFrameInParts(int nParts)
    {
        if (_frameInFlight) return;
        if (Gpus > 1) { StartFrame(); return; }                          // PollParts then reports 0 until PollFrame says done
        VrtNative.Check(_ctx, VrtNative.vrt_render_begin_parts(_ctx, ref _prm, _pixelsPin.AddrOfPinnedObject(), IdsPtr(), nParts),
                        "vrt_render_begin_parts");
        _frameInFlight = true; PartsInFlight = nParts;
    }

    /// Strip sets finished so far; part k holds the rows r with (r / tileRows) % nParts == k.
    public int PollParts()
    {
        if (!_frameInFlight || PartsInFlight == 0) return 0;
        VrtNative.Check(_ctx, VrtNative.vrt_render_poll_parts(_ctx, out int done), "vrt_render_poll_parts");
        return done;
    }

    /// True once PixelColorData holds the frame; the caller then runs UpdateRenderTexture (:461-477).
    public bool PollFrame()
    {
        if (!_frameInFlight) return false;
        VrtNative.Check(_ctx, VrtNative.vrt_render_poll(_ctx, out int done), "vrt_render_poll");
        if (done == 0) return false;
        VrtNative.Check(_ctx, VrtNative.vrt_render_end(_ctx, out VrtNative.Stats _), "vrt_render_end");
        _frameInFlight = false; PartsInFlight = 0;
        return true;
    }

    /// Single-step mode (RT_IterationMode.Single, :348): one pixel with its ray tree.  Returns the
    /// pixel colour for SetTexturePixel (:435-444) and fills RT_rec_points (:620-654) with the
    /// primary / reflection / transmission chain of sample 0.
    public Color Step(int hCord, int vCord, List<VrtNative.PathSeg> allSegments = null)
    {
        var segs = new VrtNative.PathSeg[253];                            // byte counter at :649
        VrtNative.Check(_ctx, VrtNative.vrt_trace_pixel(_ctx, ref _prm, hCord, vCord, -1, segs, segs.Length, out int n, out Color c, out int _),
                        "vrt_trace_pixel");
        RayTraceUtility.RT_rec_points.Clear();
        for (int i = 0; i < Math.Min(n, segs.Length); ++i)
        {
            if (allSegments != null) allSegments.Add(segs[i]);
            bool chain = segs[i].kind == VrtNative.RayPrimary || segs[i].kind == VrtNative.RayReflect || segs[i].kind == VrtNative.RayTransmit;
            if (segs[i].sample == 0 && chain && segs[i].depth > 0) RayTraceUtility.RT_rec_points.Add(segs[i].end);
        }
        return c;
    }

    void OnDestroy()
    {
        if (_ctx == IntPtr.Zero) return;
        if (_frameInFlight) VrtNative.vrt_render_end(_ctx, out VrtNative.Stats _);
        VrtNative.vrt_destroy(_ctx); _ctx = IntPtr.Zero;
        if (_offsetsPin.IsAllocated) _offsetsPin.Free();
        if (_pixelsPin.IsAllocated) _pixelsPin.Free();
        if (_idsPin.IsAllocated) _idsPin.Free();
    }
}
