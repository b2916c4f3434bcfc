// AbstractTracer on the GPU (Tracer/AbstractTracer.cs:5-12): installed in
// RayTraceUtility.GlobalWorld.Tracer, it serves every legacy caller that traces one ray at a time,
// including material code that recurses by hand (Material/ReflectiveMaterial.cs:38,
// Material/DielectricMaterial.cs:78-136).  One kernel launch per call: meant for the single-step
// visualisation, not for frames — those go through vrt_render / vrt_render_begin.
using System;
using UnityEngine;

public sealed class NativeTracer : AbstractTracer
{
    readonly IntPtr _ctx;
    VrtNative.Params _prm;

    public NativeTracer(IntPtr ctx, VrtNative.Params prm) { _ctx = ctx; _prm = prm; }

    public int LastHitPrimitive { get; private set; } = -1;

    public override Color TraceRay(Ray ray) => TraceRay(ray, RayTraceUtility.GlobalWorld.MaxDepth); // WhittedTracer.cs:28-31

    public override Color TraceRay(Ray ray, int depth)                                            // WhittedTracer.cs:33-104
    {
        Vector3 o = ray.origin, d = ray.direction;
        VrtNative.Check(_ctx, VrtNative.vrt_trace_ray(_ctx, ref _prm, ref o, ref d, depth, out Color c, out int id), "vrt_trace_ray");
        LastHitPrimitive = id;
        return c;
    }

    public override Color TraceRay(Ray ray, float tmin, int depth) => TraceRay(ray, depth);       // WhittedTracer.cs:106-109
}
