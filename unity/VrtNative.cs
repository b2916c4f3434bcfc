// P/Invoke binding of libvrt.so (include/vrt.h, ABI version 5) for the Unity host of
// VisualRayTrace.  Goes to Assets/Scripts/RayTracer/Native/; libvrt.so goes to
// Assets/Plugins/x86_64/.  Every struct is blittable and laid out like its C counterpart:
// Vector3 = 3 floats, Color = 4 floats, pointers as IntPtr.
//
// Not compiled in this repository's build environment (no C# toolchain there); the same calls are
// exercised through host/vrt_host.cpp and raytracing_amd/host.py.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

public static class VrtNative
{
    const string Lib = "vrt";
    const CallingConvention CC = CallingConvention.Cdecl;
    public const int AbiVersion = 5;

    public const int MatSolid = 1, MatMetal = 2, MatDielectric = 3;         // RayTraceUtility.MaterialType
    public const int MatMatte = 4, MatTransparent = 5;                      // Material/MatteMaterial.cs, Material/TransparentMaterial.cs
    public const int MaxLights = 8, MaxGpus = 16, MaxParts = 64;
    public const int LightDirectional = 0, LightPoint = 1;
    public const uint FixPrimarySky = 1, FixAoBasis = 2, FixDielectric = 4;
    public const uint OutLinearRgb = 1u << 8;   // vrt_render_device only: 3 floats per pixel before MaxToOne
    public const uint OutRgba8 = 1u << 9;       // every frame call: a Color32 per pixel, what the RGBA32 texture holds after SetPixels (RayTracerUnity.cs:472-475)
    public const int ModeAuto = 0, ModeMegakernel = 1, ModeWavefront = 2;
    public const int RayPrecheck = 0, RayPrimary = 1, RayReflect = 2, RayTransmit = 3, RayShadow = 4, RayAo = 5;

    [StructLayout(LayoutKind.Sequential)] public struct Sphere { public Vector3 center; public float radius; public int material; }
    [StructLayout(LayoutKind.Sequential)] public struct Box {
        public Vector3 center; public Vector3 axis0, axis1, axis2; public Vector3 half; public int material; }
    [StructLayout(LayoutKind.Sequential)] public struct Material { public int type; public Color color; }
    [StructLayout(LayoutKind.Sequential)] public struct Light {
        public int type; public Vector3 vec; public float ls; public Color color; public int castShadows; public Vector3 location; }
    [StructLayout(LayoutKind.Sequential)] public struct Shading {
        public float solidKa, solidKd, solidKs, solidExp;
        public float metalKa, metalKd, metalKs, metalExp, metalKr; public Color metalCr;
        public float glassKa, glassKd, glassKs, glassExp, glassEtaIn, glassEtaOut;
        public Color aoMinAmount; public float aoLs; public Color aoColor;
        public float matteKa, matteKd;
        public float transpKa, transpKd, transpKs, transpExp, transpIor, transpKr, transpKt; public Vector3 transpRayDir; }
    [StructLayout(LayoutKind.Sequential)] public struct Texture { public int width, height, wrap; public IntPtr rgba; }
    [StructLayout(LayoutKind.Sequential)] public struct SceneDesc {
        public uint nSpheres; public IntPtr spheres; public uint nBoxes; public IntPtr boxes;
        public uint nVertices; public IntPtr vertices; public uint nTriangles; public IntPtr indices; public IntPtr triMaterial;
        public uint nMaterials; public IntPtr materials; public uint nLights; public IntPtr lights; public Shading shading;
        public uint nTextures; public IntPtr textures; public IntPtr materialTexture; public IntPtr uvs; }
    [StructLayout(LayoutKind.Sequential)] public struct Camera { public Vector3 origin, p0, xAxis, yAxis; public int width, height; }
    [StructLayout(LayoutKind.Sequential)] public struct PathSeg { public Vector3 origin, end; public int kind, depth, hitPrim, sample; }
    [StructLayout(LayoutKind.Sequential)] public struct Params {
        public int spp, nSets; public IntPtr offsets; public uint seed; public int maxDepth; public float range; public uint flags; public int mode;
        public int tileIndex, tileCount, tileRows, compactRows;
        public int nRecordPixels; public IntPtr recordPixels; public int recordCap; public IntPtr recordOut; public IntPtr recordCountOut; }
    [StructLayout(LayoutKind.Sequential)] public unsafe struct Stats {
        public fixed ulong rays[6]; public ulong pixels; public float gpuMs, bvhBuildMs; public uint bvhNodes; public int modeUsed;
        public ulong nodesVisited, primsTested; }

    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_abi_version();
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_create(int device, out IntPtr ctx);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_create_multi(int nGpus, int[] devices, out IntPtr ctx);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_destroy(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_set_option(IntPtr ctx, [MarshalAs(UnmanagedType.LPStr)] string name, long value);
    // page-locked frame memory: wrap the pointer of vrt_host_alloc in a NativeArray<Color> for PixelColorData
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_host_alloc(IntPtr ctx, ulong bytes, out IntPtr buffer);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_host_free(IntPtr ctx, IntPtr buffer);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_host_register(IntPtr ctx, IntPtr buffer, ulong bytes);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_host_unregister(IntPtr ctx, IntPtr buffer);
    [DllImport(Lib, CallingConvention = CC)] public static extern IntPtr vrt_last_error(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CC)] public static extern void vrt_shading_defaults(out Shading s);
    [DllImport(Lib, CallingConvention = CC)] public static extern void vrt_params_defaults(out Params p);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_set_scene(IntPtr ctx, ref SceneDesc scene);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_set_camera(IntPtr ctx, ref Camera cam);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render(IntPtr ctx, ref Params p, [Out] Color[] rgba, [Out] int[] hitIds, out Stats stats);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_begin(IntPtr ctx, ref Params p, IntPtr rgba, IntPtr hitIds);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_begin_parts(IntPtr ctx, ref Params p, IntPtr rgba, IntPtr hitIds, int nParts);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_poll_parts(IntPtr ctx, out int partsDone);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_poll(IntPtr ctx, out int done);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_end(IntPtr ctx, out Stats stats);
    // frames that stay in GPU memory (stream = a hipStream_t or IntPtr.Zero); _noalloc allocates nothing (megakernel only)
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_device(IntPtr ctx, ref Params p, IntPtr rgbaDev, IntPtr hitIdDev, IntPtr stream);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_render_device_noalloc(IntPtr ctx, ref Params p, IntPtr rgbaDev, IntPtr hitIdDev, IntPtr stream);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_fetch_stats(IntPtr ctx, out Stats stats);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_tonemap_device(IntPtr ctx, IntPtr rgbDev, IntPtr rgbaDev, ulong nPixels, IntPtr stream);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_trace_pixel(IntPtr ctx, ref Params p, int h, int v, int sample,
        [Out] PathSeg[] segs, int cap, out int n, out Color rgba, out int hitId);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_trace_ray(IntPtr ctx, ref Params p, ref Vector3 origin, ref Vector3 direction,
        int depth, out Color rgba, out int hitId);
    [DllImport(Lib, CallingConvention = CC)] public static extern int vrt_sampler_table(int sampler, int nSamples, int nSets, float hStep, float vStep,
        ulong seed, [Out] float[] offsets);
    [DllImport(Lib, CallingConvention = CC)] public static extern uint vrt_pixel_set(uint seed, uint pixel, uint nSets);

    /// Throws with the library's message when a call reports an error.
    public static void Check(IntPtr ctx, int rc, string what)
    {
        if (rc == 0) return;
        string msg = ctx != IntPtr.Zero ? Marshal.PtrToStringAnsi(vrt_last_error(ctx)) : "no context";
        throw new InvalidOperationException($"libvrt: {what} failed with {rc}: {msg}");
    }
}
