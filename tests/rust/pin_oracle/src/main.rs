//! pin_oracle: the reference's `render_mesh_15` / `render_mesh` (src/rasterizer/render.rs:2302-2638, :1971-2264) run on `.b32scene`
//! files (layout: bonnie-32_amd/scenefile.py), printing per file the SHA-256 of `fb.pixels` and of `fb.zbuffer` and
//! `triangles_drawn`, and comparing them with the expectation record the file carries (= tests/golden/hashes.json, produced by the
//! CPU oracle oracle/b32_oracle.c).  Agreement on every file turns the oracle's "parity unpinned" into "pinned by the reference".
//!
//! UNCOMPILED SOURCE: the image this repository is built in has no Rust toolchain.  `reference` next to Cargo.toml must be (a symlink
//! to) a checkout of EBonura/bonnie-32 @ v0.1.11; the rasterizer modules are included from there with #[path], nothing is copied.
//!
//!   ln -s /path/to/bonnie-32 tests/rust/pin_oracle/reference
//!   python tools/export_scenes.py gpurun_out/scenes
//!   cargo run --release --manifest-path tests/rust/pin_oracle/Cargo.toml -- gpurun_out/scenes/*.b32scene
//!
//! The fragment-store count of the expectation record (`fragments`) is NOT checked here: the reference does not count its pixel
//! stores; frame, depth buffer and triangles_drawn are what it exposes.

#[allow(dead_code, unused_imports)]
#[path = "../reference/src/rasterizer/mod.rs"]
mod rasterizer;

/// `crate::world::Skybox` as far as render.rs names it (render.rs:83, :151: Framebuffer::render_skybox / render_stars, which this
/// harness never calls): the fields and the one method those two functions touch (src/world/geometry.rs:245-262, :321, :529, :1027).
#[allow(dead_code)]
mod world {
    use crate::rasterizer::Color;
    pub struct StarField { pub enabled: bool, pub color: Color, pub count: u16, pub size: f32, pub twinkle_speed: f32, pub seed: u32 }
    pub struct SkyboxVertex { pub pos: (f32, f32, f32), pub color: Color }
    pub struct Skybox { pub stars: StarField, pub horizon: f32 }
    impl Skybox {
        pub fn generate_mesh(&self, _camera_pos: (f32, f32, f32), _time: f32) -> (Vec<SkyboxVertex>, Vec<[usize; 3]>) { (Vec::new(), Vec::new()) }
    }
}

use rasterizer::types::OrthoProjection;
use rasterizer::{render_mesh, render_mesh_15, BlendMode, Camera, Color, Color15, Face, Framebuffer, Light, LightType, RasterSettings,
                 ShadingMode, Texture, Texture15, Vec2, Vec3, Vertex};
use sha2::{Digest, Sha256};

struct Rd<'a> { b: &'a [u8], o: usize }
impl<'a> Rd<'a> {
    fn u8(&mut self) -> u8 { let v = self.b[self.o]; self.o += 1; v }
    fn u32(&mut self) -> u32 { let v = u32::from_le_bytes(self.b[self.o..self.o + 4].try_into().unwrap()); self.o += 4; v }
    fn u64(&mut self) -> u64 { let v = u64::from_le_bytes(self.b[self.o..self.o + 8].try_into().unwrap()); self.o += 8; v }
    fn f32(&mut self) -> f32 { f32::from_bits(self.u32()) }
    fn v3(&mut self) -> Vec3 { let (x, y, z) = (self.f32(), self.f32(), self.f32()); Vec3::new(x, y, z) }
    fn bytes(&mut self, n: usize) -> &'a [u8] { let s = &self.b[self.o..self.o + n]; self.o += n; s }
}

fn blend(b: u8) -> BlendMode {                         // types.rs:1380-1388, in declaration order
    match b { 1 => BlendMode::Average, 2 => BlendMode::Add, 3 => BlendMode::Subtract, 4 => BlendMode::AddQuarter, 5 => BlendMode::Erase, _ => BlendMode::Opaque }
}
fn color(r: &mut Rd) -> Color { let (cr, cg, cb, bl) = (r.u8(), r.u8(), r.u8(), r.u8()); Color { r: cr, g: cg, b: cb, blend: blend(bl) } }
fn hex(d: &[u8]) -> String { d.iter().map(|x| format!("{:02x}", x)).collect() }

fn run(path: &str) -> bool {
    let data = std::fs::read(path).expect("read scene file");
    let mut r = Rd { b: &data, o: 0 };
    assert_eq!(r.bytes(8), b"B32SCENE", "{}: not a .b32scene file", path);
    let (version, flags) = (r.u32(), r.u32());
    assert_eq!(version, 1);
    let (w, h) = (r.u32() as usize, r.u32() as usize);
    let (nv, nf, nt, nl) = (r.u32() as usize, r.u32() as usize, r.u32() as usize, r.u32() as usize);
    let clear = color(&mut r);
    r.o = 64;
    // Camera (camera.rs:9-18): the basis vectors are data of the file (they come from sin / cos in Camera::update_basis)
    let camera = Camera { position: r.v3(), rotation_x: 0.0, rotation_y: 0.0, basis_x: r.v3(), basis_y: r.v3(), basis_z: r.v3() };
    let st: Vec<u8> = r.bytes(12).to_vec();
    let ambient = r.f32();
    let (oz, ocx, ocy) = (r.f32(), r.f32(), r.f32());
    let (fs, ff, fc) = (r.f32(), r.f32(), r.f32());
    let fog_color = color(&mut r);
    let fog = if flags & 2 != 0 { Some((fs, ff, fc, fog_color)) } else { None };
    let mut lights = Vec::with_capacity(nl);
    for _ in 0..nl {
        let ty = r.u32();
        let (position, direction) = (r.v3(), r.v3());
        let (radius, angle, intensity) = (r.f32(), r.f32(), r.f32());
        let (cr, cg, cb, enabled) = (r.u8(), r.u8(), r.u8(), r.u8());
        let light_type = match ty {                                                   // types.rs:1297-1304
            0 => LightType::Directional { direction },
            1 => LightType::Point { position, radius },
            _ => LightType::Spot { position, direction, angle, radius },
        };
        lights.push(Light { light_type, color: Color { r: cr, g: cg, b: cb, blend: BlendMode::Opaque }, intensity, enabled: enabled != 0, name: String::new() });
    }
    let settings = RasterSettings {                                                    // types.rs:1392-1428
        affine_textures: st[0] != 0, use_zbuffer: st[1] != 0,
        shading: match st[2] { 0 => ShadingMode::None, 1 => ShadingMode::Flat, _ => ShadingMode::Gouraud },
        backface_cull: st[3] != 0, backface_wireframe: st[4] != 0, lights, ambient,
        low_resolution: true, dithering: st[5] != 0, stretch_to_fill: false, wireframe_overlay: st[6] != 0,
        ortho_projection: if flags & 4 != 0 { Some(OrthoProjection { zoom: oz, center_x: ocx, center_y: ocy }) } else { None },
        use_rgb555: st[7] != 0, use_fixed_point: st[8] != 0, xray_mode: st[9] != 0,
    };
    let mut vertices = Vec::with_capacity(nv);
    for _ in 0..nv {
        let pos = r.v3();
        let uv = Vec2::new(r.f32(), r.f32());
        let normal = r.v3();
        let c = color(&mut r);
        vertices.push(Vertex { pos, uv, normal, color: c, bone_index: None });         // types.rs:947-959
    }
    let mut faces = Vec::with_capacity(nf);
    for _ in 0..nf {
        let (v0, v1, v2, tex) = (r.u32() as usize, r.u32() as usize, r.u32() as usize, r.u32());
        let (bt, bm, alpha, _pad) = (r.u8(), r.u8(), r.u8(), r.u8());
        faces.push(Face { v0, v1, v2, texture_id: if tex == 0xFFFF_FFFF { None } else { Some(tex as usize) },       // types.rs:984-1002
                          black_transparent: bt != 0, blend_mode: blend(bm), editor_alpha: alpha });
    }
    let fmt8 = flags & 1 != 0;
    let (mut tex15, mut tex8) = (Vec::new(), Vec::new());
    for _ in 0..nt {
        let (tw, th, bl, tb) = (r.u32() as usize, r.u32() as usize, r.u32(), r.u32());
        assert_eq!(tb, if fmt8 { 4 } else { 2 });
        if fmt8 {
            let px: Vec<Color> = (0..tw * th).map(|_| color(&mut r)).collect();
            tex8.push(Texture { width: tw, height: th, pixels: px, name: String::new(), blend_mode: blend(bl as u8) });             // types.rs:1058-1065
        } else {
            let px: Vec<Color15> = r.bytes(tw * th * 2).chunks_exact(2).map(|c| Color15(u16::from_le_bytes([c[0], c[1]]))).collect();
            tex15.push(Texture15 { width: tw, height: th, pixels: px, name: String::new(), blend_mode: blend(bl as u8) });         // types.rs:532-539
        }
    }
    let expect = if flags & 8 != 0 {
        let (td, _z, _fr) = (r.u32(), r.u32(), r.u64());
        Some((td, hex(r.bytes(32)), hex(r.bytes(32))))
    } else { None };
    assert_eq!(r.o, data.len(), "{}: trailing bytes", path);

    let mut fb = Framebuffer::new(w, h);                                               // render.rs:18-25
    fb.clear(clear);                                                                    // render.rs:36-45
    let tm = if fmt8 { render_mesh(&mut fb, &vertices, &faces, &tex8, &camera, &settings) }
             else { render_mesh_15(&mut fb, &vertices, &faces, &tex15, &camera, &settings, fog) };
    let frame = hex(&Sha256::digest(&fb.pixels));
    let zbytes: Vec<u8> = fb.zbuffer.iter().flat_map(|z| z.to_le_bytes()).collect();
    let zsha = hex(&Sha256::digest(&zbytes));
    let verdict = match &expect {
        Some((td, f, z)) => if *td == tm.triangles_drawn && *f == frame && *z == zsha { "PINNED" } else { "MISMATCH" },
        None => "no expectation in file",
    };
    println!("{path}: sha256 {frame} zbuffer_sha256 {zsha} triangles_drawn {} -- {verdict}", tm.triangles_drawn);
    if let Some((td, f, z)) = &expect {
        if verdict == "MISMATCH" { println!("    oracle says: sha256 {f} zbuffer_sha256 {z} triangles_drawn {td}"); }
    }
    verdict != "MISMATCH"
}

fn main() {
    let files: Vec<String> = std::env::args().skip(1).collect();
    if files.is_empty() { eprintln!("usage: pin_oracle <scene.b32scene> ..."); std::process::exit(2); }
    let bad = files.iter().filter(|f| !run(f)).count();
    println!("{} file(s), {} mismatch(es)", files.len(), bad);
    std::process::exit(if bad == 0 { 0 } else { 1 });
}
