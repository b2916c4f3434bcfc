// Flattens the Unity scene the reference traces through Physics.Raycast into a vrt_scene_desc:
// every active collider that is not on the IgnoreRayCast layer with the material of its
// MeshRenderer (what WhittedTracer.TraceRay looks up per hit, Tracer/WhittedTracer.cs:43-88), and
// every Light the way the WorldInformation constructor reads it (Util/RayTraceUtility.cs:493-527).
// Inputs are copied by vrt_set_scene, so the pinned arrays are released before Upload returns.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using UnityEngine;

public sealed class VrtSceneMarshaller
{
    readonly List<VrtNative.Sphere> _spheres = new List<VrtNative.Sphere>();
    readonly List<VrtNative.Box> _boxes = new List<VrtNative.Box>();
    readonly List<Vector3> _vertices = new List<Vector3>();
    readonly List<Vector2> _uvs = new List<Vector2>();
    readonly List<uint> _indices = new List<uint>();
    readonly List<int> _triMaterial = new List<int>();
    readonly List<VrtNative.Material> _materials = new List<VrtNative.Material>();
    readonly List<int> _materialTexture = new List<int>();
    readonly List<Texture2D> _textures = new List<Texture2D>();
    readonly List<VrtNative.Light> _lights = new List<VrtNative.Light>();
    readonly Dictionary<Material, int> _materialIndex = new Dictionary<Material, int>();

    /// Primitive id reported by the library -> the collider it came from (ids run over spheres,
    /// then boxes, then triangles; every triangle of a mesh maps to its MeshCollider).
    public readonly List<Collider> PrimitiveOwner = new List<Collider>();

    int MaterialOf(Collider c)
    {
        var renderer = c.GetComponent<MeshRenderer>();
        // WhittedTracer.cs:49 reads renderer.material, which on first access replaces the renderer's
        // material by a private copy ("<name> (Instance)") and returns that.  sharedMaterial returns
        // whatever the renderer holds now without copying: the asset while nothing has touched
        // .material, the private copy afterwards, so a colour set through renderer.material.color
        // by other scripts is seen here as well.  The values marshalled are the same; what differs
        // is that collecting the scene does not instantiate a material per renderer, and that
        // renderers sharing an asset share one entry of the material table.  Per-renderer
        // MaterialPropertyBlock overrides are invisible to both properties.
        Material mat = renderer != null ? renderer.sharedMaterial : null;
        if (mat == null) mat = new Material(Shader.Find("Standard")) { name = "Missing", color = Color.red }; // WhittedTracer.cs:52
        if (_materialIndex.TryGetValue(mat, out int idx)) return idx;
        idx = _materials.Count;
        _materials.Add(new VrtNative.Material { type = (int)RayTraceUtility.DetermineMaterialType(mat), color = mat.color });
        int tex = -1;
        if (mat.mainTexture is Texture2D t2d && t2d.isReadable) { tex = _textures.Count; _textures.Add(t2d); } // :55-63
        _materialTexture.Add(tex);
        _materialIndex[mat] = idx;
        return idx;
    }

    public void Collect(int layerMask)
    {
        var spheres = new List<Collider>(); var boxes = new List<Collider>(); var meshes = new List<MeshCollider>();
        foreach (var c in UnityEngine.Object.FindObjectsOfType<Collider>())
        {
            if (!c.enabled || !c.gameObject.activeInHierarchy) continue;
            if ((layerMask & (1 << c.gameObject.layer)) == 0) continue;   // RayTraceUtility.LayerMask = ~(1 << 9)
            if (c is SphereCollider) spheres.Add(c); else if (c is BoxCollider) boxes.Add(c); else if (c is MeshCollider m) meshes.Add(m);
        }
        foreach (SphereCollider s in spheres)
        {
            Vector3 sc = s.transform.lossyScale;
            float r = s.radius * Mathf.Max(Mathf.Abs(sc.x), Mathf.Abs(sc.y), Mathf.Abs(sc.z));
            _spheres.Add(new VrtNative.Sphere { center = s.transform.TransformPoint(s.center), radius = r, material = MaterialOf(s) });
            PrimitiveOwner.Add(s);
        }
        foreach (BoxCollider b in boxes)
        {
            Transform t = b.transform; Vector3 sc = t.lossyScale;
            _boxes.Add(new VrtNative.Box {
                center = t.TransformPoint(b.center), axis0 = t.right, axis1 = t.up, axis2 = t.forward,
                half = new Vector3(Mathf.Abs(b.size.x * sc.x), Mathf.Abs(b.size.y * sc.y), Mathf.Abs(b.size.z * sc.z)) * 0.5f,
                material = MaterialOf(b) });
            PrimitiveOwner.Add(b);
        }
        foreach (MeshCollider m in meshes)
        {
            Mesh mesh = m.sharedMesh; if (mesh == null) continue;
            int mat = MaterialOf(m); uint first = (uint)_vertices.Count;
            Vector3[] v = mesh.vertices; Vector2[] uv = mesh.uv; int[] tri = mesh.triangles;
            for (int i = 0; i < v.Length; ++i) { _vertices.Add(m.transform.TransformPoint(v[i])); _uvs.Add(i < uv.Length ? uv[i] : Vector2.zero); }
            for (int i = 0; i + 2 < tri.Length; i += 3)
            {
                _indices.Add(first + (uint)tri[i]); _indices.Add(first + (uint)tri[i + 1]); _indices.Add(first + (uint)tri[i + 2]);
                _triMaterial.Add(mat); PrimitiveOwner.Add(m);
            }
        }
        // Util/RayTraceUtility.cs:493-527 (FindObjectsOfTypeAll there also returns inactive lights)
        foreach (Light l in Resources.FindObjectsOfTypeAll(typeof(Light)) as Light[])
        {
            if (l.type == LightType.Directional)
            {
                Vector3 dir = new Vector3(0, 0, 1), e = l.transform.rotation.eulerAngles;
                if (e.x != 0f) dir = Quaternion.AngleAxis(e.x, Vector3.right) * dir;
                if (e.y != 0f) dir = Quaternion.AngleAxis(e.y, Vector3.up) * dir;
                if (e.z != 0f) dir = Quaternion.AngleAxis(e.z, Vector3.forward) * dir;
                // DirectionalLight: colour forced white, _location never assigned, shadows off
                _lights.Add(new VrtNative.Light { type = VrtNative.LightDirectional, vec = dir, ls = l.intensity, color = Color.white,
                                                  castShadows = 0, location = Vector3.zero });
            }
            else if (l.type == LightType.Point)
                _lights.Add(new VrtNative.Light { type = VrtNative.LightPoint, vec = l.transform.position, ls = l.intensity, color = l.color,
                                                  castShadows = 1, location = Vector3.zero });
        }
    }

    public void Upload(IntPtr ctx)
    {
        var pins = new List<GCHandle>();
        IntPtr Pin(Array a) { if (a == null || a.Length == 0) return IntPtr.Zero; var h = GCHandle.Alloc(a, GCHandleType.Pinned); pins.Add(h); return h.AddrOfPinnedObject(); }
        try
        {
            var desc = new VrtNative.SceneDesc();
            desc.nSpheres = (uint)_spheres.Count; desc.spheres = Pin(_spheres.ToArray());
            desc.nBoxes = (uint)_boxes.Count; desc.boxes = Pin(_boxes.ToArray());
            desc.nVertices = (uint)_vertices.Count; desc.vertices = Pin(_vertices.ToArray());
            desc.nTriangles = (uint)(_indices.Count / 3); desc.indices = Pin(_indices.ToArray()); desc.triMaterial = Pin(_triMaterial.ToArray());
            desc.nMaterials = (uint)_materials.Count; desc.materials = Pin(_materials.ToArray());
            // all of them: more than VrtNative.MaxLights is reported by vrt_set_scene (VRT_ERR_CAPACITY), not cut silently
            desc.nLights = (uint)_lights.Count; desc.lights = Pin(_lights.ToArray());
            VrtNative.vrt_shading_defaults(out desc.shading);
            // the sliders of the options canvas write these statics (OptionsCanvas/CanvasSettings.cs:276-325)
            desc.shading.metalKa = RayTraceUtility.Metal_KA; desc.shading.metalKd = RayTraceUtility.Metal_KD;
            desc.shading.metalKs = RayTraceUtility.Metal_KS; desc.shading.metalExp = RayTraceUtility.Metal_EXP;
            desc.shading.metalKr = RayTraceUtility.Metal_KR; desc.shading.glassKs = RayTraceUtility.Dielectric_KS;
            desc.shading.glassEtaIn = RayTraceUtility.Dielectric_EtaIN; desc.shading.glassEtaOut = RayTraceUtility.Dielectric_EtaOUT;
            var tex = new VrtNative.Texture[_textures.Count];
            for (int i = 0; i < tex.Length; ++i)
            {
                Color[] px = _textures[i].GetPixels();   // row 0 = bottom row, as GetPixel indexes
                tex[i] = new VrtNative.Texture { width = _textures[i].width, height = _textures[i].height,
                                                 wrap = _textures[i].wrapMode == TextureWrapMode.Clamp ? 1 : 0, rgba = Pin(px) };
            }
            if (tex.Length > 0)
            {
                desc.nTextures = (uint)tex.Length; desc.textures = Pin(tex);
                desc.materialTexture = Pin(_materialTexture.ToArray()); desc.uvs = Pin(_uvs.ToArray());
            }
            VrtNative.Check(ctx, VrtNative.vrt_set_scene(ctx, ref desc), "vrt_set_scene");
        }
        finally { foreach (var h in pins) h.Free(); }
    }
}
