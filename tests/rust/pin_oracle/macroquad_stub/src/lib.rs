//! Stand-in for the one macroquad item the reference's rasterizer uses: `macroquad::prelude::get_time` (render.rs:4), the clock behind
//! RasterTimings.  The real function needs a macroquad window context; a monotonic clock has the same meaning for the timings and no
//! effect on any pixel.
pub mod prelude {
    pub fn get_time() -> f64 {
        use std::time::Instant;
        static START: std::sync::OnceLock<Instant> = std::sync::OnceLock::new();
        START.get_or_init(Instant::now).elapsed().as_secs_f64()
    }
}
