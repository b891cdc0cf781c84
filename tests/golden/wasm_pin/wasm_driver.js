// Runs the reference's own compiled code -- render_mesh of docs/bonnie-engine.wasm, an older build of the crate (see README.md here) --
// on a .b32scene file and writes the framebuffer it produces.  Nothing of the rasterizer is re-implemented here: this file only lays
// the inputs out in the module's linear memory the way that build's rustc laid out its structs (offsets read from the module's own
// code, listed in README.md) and calls the module's functions.
//   node --experimental-wasm-anyref --experimental-wasm-bulk-memory wasm_driver.js patched.wasm in.b32scene out.rgba
//   node ... wasm_driver.js patched.wasm --acosf in.f32 out.f32
// patched.wasm = the reference's file with three extra EXPORT entries (wasm_tools.py adds them; no code byte changes).
'use strict';
const fs = require('fs');
const [wasmPath, scenePath, outPath] = process.argv.slice(2);
const mod = new WebAssembly.Module(fs.readFileSync(wasmPath));
const called = [];
const imports = {};
for (const imp of WebAssembly.Module.imports(mod)) {
  imports[imp.module] = imports[imp.module] || {};
  if (imp.kind !== 'function') continue;
  // the only import render_mesh reaches is the clock of its RasterTimings (miniquad's `now`); anything else is an error
  imports[imp.module][imp.name] = imp.name === 'now' ? () => 0.0 : () => { throw new Error('unexpected import call: ' + imp.name); };
}
const ex = new WebAssembly.Instance(mod, imports).exports;
const mem = () => new DataView(ex.memory.buffer);
const alloc = n => { const p = ex.pin_malloc(Math.max(n, 4)); if (!p) throw new Error('malloc'); new Uint8Array(ex.memory.buffer, p, n).fill(0); return p; };

// macroquad's get_time() (called for the timings) checks that a context exists and that it is called from the thread that created
// it; render_mesh needs nothing else of the context.  The three statics it tests, at the addresses this binary's code reads:
{
  const d = mem();
  d.setBigUint64(1198272, 1n, true);   // THREAD_ID: Some(1)
  d.setUint8(1198264, 1);              // thread-local "current thread id": initialised ...
  d.setBigUint64(1198256, 1n, true);   // ... to 1
  d.setBigUint64(1195560, 0n, true);   // CONTEXT: not None (None is encoded as 2)
  d.setFloat64(1197984, 0.0, true);    // its start time
}

// ---- mode 2: the module's acosf (the `libm` crate's, what f32::acos is on wasm32) over a file of f32 inputs
if (scenePath === '--acosf') {
  const inPath = outPath, outP = process.argv[5];
  const raw = fs.readFileSync(inPath);
  const xs = new Float32Array(raw.buffer, raw.byteOffset, raw.byteLength / 4), ys = new Float32Array(xs.length);
  for (let i = 0; i < xs.length; ++i) ys[i] = ex.pin_acosf(xs[i]);
  fs.writeFileSync(outP, Buffer.from(ys.buffer));
  process.exit(0);
}

// ---- .b32scene (bonnie-32_amd/scenefile.py)
const f = fs.readFileSync(scenePath);
const fd = new DataView(f.buffer, f.byteOffset, f.byteLength);
if (f.toString('latin1', 0, 8) !== 'B32SCENE') throw new Error('not a .b32scene');
const [ver, flags, W, H, nv, nf, nt, nl] = [8, 12, 16, 20, 24, 28, 32, 36].map(o => fd.getUint32(o, true));
if (ver !== 1 || !(flags & 1)) throw new Error('need a version-1 scene of the 8-bit-colour path');
if (flags & 2) throw new Error('fog: this build of render_mesh has no fog argument');
const clear = [f[40], f[41], f[42]];
let o = 64;
const camOff = o; o += 48;
const s8 = f.slice(o, o + 12); o += 12;
const ambient = fd.getFloat32(o, true), oz = fd.getFloat32(o + 4, true), ocx = fd.getFloat32(o + 8, true), ocy = fd.getFloat32(o + 12, true); o += 16;
o += 16;                                // fog record
const lightsOff = o; o += nl * 44;
const vertsOff = o; o += nv * 36;
const facesOff = o; o += nf * 20;
if (s8[7] || s8[8] || s8[9]) throw new Error('use_rgb555 / use_fixed_point / xray_mode do not exist in this build');

// ---- the module's structs
// Framebuffer { pixels: Vec<u8> @0 (cap, ptr, len), zbuffer: Vec<f32> @12, width @24, height @28 } -- built by the module itself
const fb = alloc(32);
ex.pin_fb_new(fb, W, H);
{
  const d = mem(), px = d.getUint32(fb + 4, true), n = d.getUint32(fb + 8, true);
  if (n !== W * H * 4) throw new Error('Framebuffer::new');
  const p = new Uint8Array(ex.memory.buffer, px, n);              // Framebuffer::clear(color): r, g, b, 255 per pixel
  for (let i = 0; i < n; i += 4) { p[i] = clear[0]; p[i + 1] = clear[1]; p[i + 2] = clear[2]; p[i + 3] = 255; }
}
// Vertex (44 B) { bone_index: Option<usize> @0, color: Color @8, pos @12, uv @24, normal @32 };  Color { blend @0, r @1, g @2, b @3 }
const verts = alloc(nv * 44);
{
  const d = mem();
  for (let i = 0; i < nv; ++i) {
    const s = vertsOff + i * 36, t = verts + i * 44;
    for (let k = 0; k < 3; ++k) d.setFloat32(t + 12 + 4 * k, fd.getFloat32(s + 4 * k, true), true);
    for (let k = 0; k < 2; ++k) d.setFloat32(t + 24 + 4 * k, fd.getFloat32(s + 12 + 4 * k, true), true);
    for (let k = 0; k < 3; ++k) d.setFloat32(t + 32 + 4 * k, fd.getFloat32(s + 20 + 4 * k, true), true);
    d.setUint8(t + 8, f[s + 35]); d.setUint8(t + 9, f[s + 32]); d.setUint8(t + 10, f[s + 33]); d.setUint8(t + 11, f[s + 34]);
  }
}
// Face (20 B) { texture_id: Option<usize> @0 (tag, value), v0 @8, v1 @12, v2 @16 }
const faces = alloc(nf * 20);
{
  const d = mem();
  for (let i = 0; i < nf; ++i) {
    const s = facesOff + i * 20, t = faces + i * 20;
    const tex = fd.getUint32(s + 12, true);
    if (f[s + 18] !== 255) throw new Error('editor_alpha does not exist in this build');
    d.setUint32(t, tex === 0xFFFFFFFF ? 0 : 1, true); d.setUint32(t + 4, tex === 0xFFFFFFFF ? 0 : tex, true);
    for (let k = 0; k < 3; ++k) d.setUint32(t + 8 + 4 * k, fd.getUint32(s + 4 * k, true), true);
  }
}
// Texture (32 B) { pixels: Vec<Color> @0 (cap, ptr, len), name: String @12, width @24, height @28 }
const texs = alloc(Math.max(nt, 1) * 32);
for (let i = 0; i < nt; ++i) {
  const tw = fd.getUint32(o, true), th = fd.getUint32(o + 4, true), tb = fd.getUint32(o + 12, true); o += 16;
  if (tb !== 4) throw new Error('texel size');
  const n = tw * th, px = alloc(n * 4);
  const d = mem();
  for (let k = 0; k < n; ++k) { d.setUint8(px + 4 * k, f[o + 4 * k + 3]); d.setUint8(px + 4 * k + 1, f[o + 4 * k]); d.setUint8(px + 4 * k + 2, f[o + 4 * k + 1]); d.setUint8(px + 4 * k + 3, f[o + 4 * k + 2]); }
  o += n * 4;
  const t = texs + i * 32;
  d.setUint32(t, n, true); d.setUint32(t + 4, px, true); d.setUint32(t + 8, n, true);
  d.setUint32(t + 12, 0, true); d.setUint32(t + 16, 1, true); d.setUint32(t + 20, 0, true);
  d.setUint32(t + 24, tw, true); d.setUint32(t + 28, th, true);
}
// Camera (56 B) { position @0, rotation_x @12, rotation_y @16, basis_x @20, basis_y @32, basis_z @44 }
const cam = alloc(56);
{
  const d = mem();
  for (let k = 0; k < 3; ++k) d.setFloat32(cam + 4 * k, fd.getFloat32(camOff + 4 * k, true), true);
  for (let k = 0; k < 9; ++k) d.setFloat32(cam + 20 + 4 * k, fd.getFloat32(camOff + 12 + 4 * k, true), true);
}
// Light (60 B) { light_type @0: tag 0 Directional { direction @4 } / 1 Point { position @4, radius @16 } /
//   2 Spot { position @4, direction @16, angle @28, radius @32 }, name: String @36, color @48, intensity @52, enabled @56 }
const lights = alloc(Math.max(nl, 1) * 60);
{
  const d = mem();
  for (let i = 0; i < nl; ++i) {
    const s = lightsOff + i * 44, t = lights + i * 60;
    const ty = fd.getUint32(s, true), g = k => fd.getFloat32(s + 4 + 4 * k, true);   // position[3], direction[3], radius, angle, intensity
    d.setUint32(t, ty, true);
    if (ty === 0) for (let k = 0; k < 3; ++k) d.setFloat32(t + 4 + 4 * k, g(3 + k), true);
    else if (ty === 1) { for (let k = 0; k < 3; ++k) d.setFloat32(t + 4 + 4 * k, g(k), true); d.setFloat32(t + 16, g(6), true); }
    else if (ty === 2) { for (let k = 0; k < 3; ++k) { d.setFloat32(t + 4 + 4 * k, g(k), true); d.setFloat32(t + 16 + 4 * k, g(3 + k), true); } d.setFloat32(t + 28, g(7), true); d.setFloat32(t + 32, g(6), true); }
    else throw new Error('light type');
    d.setUint32(t + 36, 0, true); d.setUint32(t + 40, 1, true); d.setUint32(t + 44, 0, true);
    d.setUint8(t + 48, 0); d.setUint8(t + 49, f[s + 40]); d.setUint8(t + 50, f[s + 41]); d.setUint8(t + 51, f[s + 42]);
    d.setFloat32(t + 52, g(8), true); d.setUint8(t + 56, f[s + 43]);
  }
}
// RasterSettings (44 B) { ortho_projection: Option<OrthoProjection> @0 (tag, zoom, center_x, center_y), lights: Vec<Light> @16, ambient @28,
//   affine_textures @32, vertex_snap @33, use_zbuffer @34, backface_cull @35, backface_wireframe @36, low_resolution @37, dithering @38,
//   stretch_to_fill @39, wireframe_overlay @40, shading @41 }
const st = alloc(44);
{
  const d = mem();
  d.setUint32(st, (flags & 4) ? 1 : 0, true); d.setFloat32(st + 4, oz, true); d.setFloat32(st + 8, ocx, true); d.setFloat32(st + 12, ocy, true);
  d.setUint32(st + 16, nl, true); d.setUint32(st + 20, nl ? lights : 4, true); d.setUint32(st + 24, nl, true);
  d.setFloat32(st + 28, ambient, true);
  d.setUint8(st + 32, s8[0]); d.setUint8(st + 33, 0); d.setUint8(st + 34, s8[1]); d.setUint8(st + 35, s8[3]); d.setUint8(st + 36, s8[4]);
  d.setUint8(st + 37, 0); d.setUint8(st + 38, s8[5]); d.setUint8(st + 39, 0); d.setUint8(st + 40, s8[6]); d.setUint8(st + 41, s8[2]);
}
// render_mesh(out: *mut RasterTimings, fb, vertices (ptr, len), faces (ptr, len), textures (ptr, len), camera, settings)
const tm = alloc(32);
ex.pin_render_mesh(tm, fb, verts, nv, faces, nf, texs, nt, cam, st);
const d = mem();
const px = d.getUint32(fb + 4, true), zb = d.getUint32(fb + 16, true);
fs.writeFileSync(outPath, Buffer.from(new Uint8Array(ex.memory.buffer, px, W * H * 4)));
fs.writeFileSync(outPath + '.z', Buffer.from(new Uint8Array(ex.memory.buffer, zb, W * H * 4)));
const words = []; for (let k = 0; k < 6; ++k) words.push(d.getUint32(tm + 4 * k, true));
console.log(JSON.stringify({ width: W, height: H, timings_words: words }));
