//! Headless steps/s of rapier3d's own CPU `parallel` path on the BASELINE.json scenes — written against the API of the
//! reference tree (`PhysicsWorld`, /root/reference/src/pipeline/physics_world.rs:61-157; scene formulas from
//! /root/reference/examples3d/{b3d_many_pyramids,b3d_large_pyramid,b3d_joint_grid}.rs).  Not compiled in this repository's image
//! (no Rust toolchain); kept as source so a maintainer can produce the true reference number on the GPU box's host CPU.
//!
//!   cargo run --release -- <many_pyramids|large_pyramid|joint_grid|pyramid10|many_pyramids_c4|reference_pile> [warmup_steps] [timed_steps]
//!   cargo run --release -- <scene> --dump <dir>     # body states after 1, 10, 100, 1000 steps -> <dir>/<scene>_s<steps>.rpdump
//!
//! `--dump` closes SURVEY §8(c) on any machine with cargo: copy the .rpdump files to tests/golden/reference/ and
//! tests/test_reference_dump.py compares the oracle with them (1e-4 relative on positions, the BASELINE.json tolerance).
//!   cargo run --release -- <scene> --trace <file> [steps]   # per-step trace -> tests/reference_trace.py diff <file> <oracle trace>
//!
//! `--trace` writes what is needed to bisect a divergence between this crate and the oracle in ONE run: for every step the state
//! hash, the number of sleeping bodies, of contact pairs holding a solver contact and of solver contacts, followed by that step's
//! transitions — `S body` / `W body` (fell asleep / woke up) and `B c1 c2` / `E c1 c2` (the pair gained its first / lost its last
//! solver contact), all by arena index.  tests/reference_trace.py writes the same file from the oracle
//! (tests/golden/reference_pile_s120.rptrace is committed) and reports the first step at which the two differ, and how.
//! Dump file format (little endian): 8 bytes "RPDUMP1\0", u32 body count, u32 steps, then per body in handle-index order 13 f32:
//! translation xyz, rotation xyzw, linvel xyz, angvel xyz; the last 8 bytes are the FNV-1a state hash of
//! crates/rapier3d/tests/simd_backend_determinism.rs:36-57 over the same floats.

use rapier3d::prelude::*;
use std::time::Instant;

/// Cube centres of one 2-D pyramid of `base` cubes (half extent `e`) relative to its left end: row r holds base - r cubes.
fn pyramid_cells(base: i32, e: f32) -> impl Iterator<Item = (f32, f32)> {
    (0..base).flat_map(move |row| (row..base).map(move |col| ((row as f32 + 1.0) * e + 2.0 * (col - row) as f32 * e, (2.0 * row as f32 + 1.0) * e)))
}

fn add_cube(world: &mut PhysicsWorld, at: Vector, e: f32) {
    world.insert(RigidBodyBuilder::dynamic().translation(at).can_sleep(false), ColliderBuilder::cuboid(e, e, e).density(100.0));
}

/// b3d_many_pyramids.rs:36-64 (rows = cols = 14) and the one-pyramid plumbing case (rows = cols = 1)
fn many_pyramids(rows: i32, cols: i32) -> PhysicsWorld {
    let mut world = PhysicsWorld::new();
    world.gravity = Vector::new(0.0, -10.0, 0.0);
    let (base_count, extent) = (10i32, 0.5f32);
    let ground_extent = extent * cols as f32 * (base_count as f32 + 1.0);
    world.insert(
        RigidBodyBuilder::fixed().translation(Vector::new(0.0, -1.0, 0.0)),
        ColliderBuilder::cuboid(ground_extent, 1.0, ground_extent),
    );
    let base_width = 2.0 * extent * base_count as f32;
    let mut base_z = -ground_extent + 2.0 * extent;
    let delta_z = if rows > 1 { 2.0 * (ground_extent - 2.0 * extent) / (rows as f32 - 1.0) } else { 0.0 };
    for _ in 0..rows {
        for j in 0..cols {
            let center_x = -ground_extent + j as f32 * (base_width + 2.0 * extent) + 2.0 * extent;
            for (dx, y) in pyramid_cells(base_count, extent) {
                add_cube(&mut world, Vector::new(dx + center_x - 0.5, y, base_z), extent);
            }
        }
        base_z += delta_z;
    }
    world
}

/// b3d_large_pyramid.rs:15-36
fn large_pyramid(base_count: i32) -> PhysicsWorld {
    let mut world = PhysicsWorld::new();
    world.gravity = Vector::new(0.0, -10.0, 0.0);
    let extent = 0.5f32;
    world.insert(
        RigidBodyBuilder::fixed().translation(Vector::new(0.0, -1.0, 0.0)),
        ColliderBuilder::cuboid(400.0, 1.0, 400.0),
    );
    for (dx, y) in pyramid_cells(base_count, extent) {
        add_cube(&mut world, Vector::new(dx - 100.0, y, 0.0), extent);
    }
    world
}

/// b3d_joint_grid.rs:17-53
fn joint_grid(n: usize) -> PhysicsWorld {
    let mut world = PhysicsWorld::new();
    world.gravity = Vector::new(0.0, -10.0, 0.0);
    let mut handles = vec![RigidBodyHandle::invalid(); n * n];
    for k in 0..n {
        for i in 0..n {
            let builder = if i == 0 { RigidBodyBuilder::fixed() } else { RigidBodyBuilder::dynamic().can_sleep(false) };
            let (h, _) = world.insert(
                builder.translation(Vector::new(k as f32, -(i as f32), 0.0)),
                ColliderBuilder::ball(0.4).density(1.0),
            );
            handles[k * n + i] = h;
            if i > 0 {
                let j = SphericalJointBuilder::new().local_anchor1(Vector::new(0.0, -0.5, 0.0)).local_anchor2(Vector::new(0.0, 0.5, 0.0));
                world.insert_impulse_joint(handles[k * n + i - 1], h, j);
            }
            if k > 0 {
                let j = SphericalJointBuilder::new().local_anchor1(Vector::new(0.5, 0.0, 0.0)).local_anchor2(Vector::new(-0.5, 0.0, 0.0));
                world.insert_impulse_joint(handles[(k - 1) * n + i], h, j);
            }
        }
    }
    world
}

/// simd_backend_determinism.rs:61-139: jittered 12 x 3 x 12 pile + a 4-ball spherical-joint chain (default sleeping, g = 9.81)
fn reference_pile() -> PhysicsWorld {
    let mut world = PhysicsWorld::new();
    world.gravity = Vector::Y * -9.81;
    world.insert(RigidBodyBuilder::fixed().translation(Vector::new(0.0, -0.5, 0.0)), ColliderBuilder::cuboid(20.0, 0.5, 20.0));
    for i in 0..12 {
        for j in 0..3 {
            for k in 0..12 {
                let jitter = (i as f32 * 0.013 + k as f32 * 0.017) % 0.05;
                world.insert(
                    RigidBodyBuilder::dynamic().translation(Vector::new(i as f32 * 1.05 - 6.0 + jitter, j as f32 * 1.05 + 0.55, k as f32 * 1.05 - 6.0 - jitter)),
                    ColliderBuilder::cuboid(0.5, 0.5, 0.5),
                );
            }
        }
    }
    let anchor = world.insert_body(RigidBodyBuilder::fixed().translation(Vector::new(0.0, 8.0, 0.0))); // no collider, as in the test
    let mut prev = anchor;
    for i in 0..4 {
        let (rb, _) = world.insert(RigidBodyBuilder::dynamic().translation(Vector::new(0.6 * (i + 1) as f32, 8.0, 0.0)), ColliderBuilder::ball(0.25));
        world.insert_impulse_joint(prev, rb, SphericalJointBuilder::new().local_anchor1(Vector::X * 0.3).local_anchor2(Vector::X * -0.3));
        prev = rb;
    }
    world
}

/// examples3d/b3d_large_world.rs: `grid` x `grid` parentless fixed cuboids (half extents 5, 0.25, 5); the spheres are dropped by the
/// timed loop (`large_world_drop`), one every five steps — tools/large_world.py runs the same script on the device
fn large_world(grid: i32) -> PhysicsWorld {
    let mut world = PhysicsWorld::new();
    world.gravity = Vector::new(0.0, -10.0, 0.0);
    let cell = 10.0f32;
    let half_span = 0.5 * cell * grid as f32;
    for i in 0..grid {
        for j in 0..grid {
            let at = Vector::new(-half_span + (i as f32 + 0.5) * cell, 0.0, -half_span + (j as f32 + 0.5) * cell);
            world.insert_collider(ColliderBuilder::cuboid(0.5 * cell, 0.25, 0.5 * cell).translation(at), None);
        }
    }
    world
}
fn large_world_drop(world: &mut PhysicsWorld, idx: i32, grid: i32, spheres: i32) {
    let mut side = 1;
    while side * side < spheres {
        side += 1;
    }
    let half_span = 0.5 * 10.0 * grid as f32;
    let inset = 0.1 * 2.0 * half_span;
    let usable = 2.0 * half_span - 2.0 * inset;
    let x = -half_span + inset + ((idx % side) as f32 + 0.5) * (usable / side as f32);
    let z = -half_span + inset + ((idx / side) as f32 + 0.5) * (usable / side as f32);
    world.insert(RigidBodyBuilder::dynamic().translation(Vector::new(x, 1.5, z)), ColliderBuilder::ball(0.5));
}

/// Per-step trace (see the file header); `steps` steps of `world`.
fn trace(world: &mut PhysicsWorld, scene: &str, path: &std::path::Path, steps: u32) {
    use std::collections::BTreeSet;
    use std::fmt::Write as _;
    let mut handles: Vec<_> = world.bodies.iter().map(|(h, _)| h).collect();
    handles.sort_by_key(|h| h.into_raw_parts().0);
    let mut out = format!("RPTRACE1 {scene} {} {steps}\n", handles.len());
    let mut asleep: BTreeSet<u32> = BTreeSet::new();
    let mut touching: BTreeSet<(u32, u32)> = BTreeSet::new();
    for step in 1..=steps {
        world.step();
        let mut hash: u64 = 0xcbf29ce484222325;
        let mut now_asleep = BTreeSet::new();
        for h in &handles {
            let rb = &world.bodies[*h];
            if rb.is_sleeping() {
                now_asleep.insert(h.into_raw_parts().0);
            }
            let vals = rb.translation().to_array().into_iter().chain(rb.rotation().to_array()).chain(rb.linvel().to_array()).chain(rb.angvel().to_array());
            for v in vals {
                for b in v.to_bits().to_le_bytes() {
                    hash ^= b as u64;
                    hash = hash.wrapping_mul(0x100000001b3);
                }
            }
        }
        let mut now_touching = BTreeSet::new();
        let mut contacts = 0usize;
        for pair in world.contact_pairs() {
            let n: usize = pair.solver_manifolds().iter().map(|m| m.data.solver_contacts.len()).sum();
            if n > 0 {
                let (a, b) = (pair.collider1.into_raw_parts().0, pair.collider2.into_raw_parts().0);
                now_touching.insert((a.min(b), a.max(b)));
                contacts += n;
            }
        }
        writeln!(out, "step {step} hash {hash:016x} asleep {} touching {} contacts {contacts}", now_asleep.len(), now_touching.len()).unwrap();
        for b in now_asleep.difference(&asleep) {
            writeln!(out, "S {b}").unwrap();
        }
        for b in asleep.difference(&now_asleep) {
            writeln!(out, "W {b}").unwrap();
        }
        for (a, b) in now_touching.difference(&touching) {
            writeln!(out, "B {a} {b}").unwrap();
        }
        for (a, b) in touching.difference(&now_touching) {
            writeln!(out, "E {a} {b}").unwrap();
        }
        asleep = now_asleep;
        touching = now_touching;
    }
    std::fs::write(path, out).expect("write trace");
    println!("{{\"trace\": \"{}\", \"steps\": {steps}}}", path.display());
}

/// Body states in handle-index order + the reference's FNV-1a state hash (see the file header for the format).
fn dump(world: &PhysicsWorld, path: &std::path::Path, steps: u32) {
    let mut handles: Vec<_> = world.bodies.iter().map(|(h, _)| h).collect();
    handles.sort_by_key(|h| h.into_raw_parts().0);
    let mut out: Vec<u8> = b"RPDUMP1\0".to_vec();
    out.extend_from_slice(&(handles.len() as u32).to_le_bytes());
    out.extend_from_slice(&steps.to_le_bytes());
    let mut hash: u64 = 0xcbf29ce484222325;
    for h in handles {
        let rb = &world.bodies[h];
        let vals = rb.translation().to_array().into_iter().chain(rb.rotation().to_array()).chain(rb.linvel().to_array()).chain(rb.angvel().to_array());
        for v in vals {
            for b in v.to_bits().to_le_bytes() {
                hash ^= b as u64;
                hash = hash.wrapping_mul(0x100000001b3);
                out.push(b);
            }
        }
    }
    out.extend_from_slice(&hash.to_le_bytes());
    std::fs::write(path, out).expect("write dump");
    println!("{{\"dump\": \"{}\", \"steps\": {steps}, \"state_hash\": \"{hash:#018x}\"}}", path.display());
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    let scene = args.get(1).map(String::as_str).unwrap_or("many_pyramids");
    let mut world = match scene {
        "many_pyramids" => many_pyramids(14, 14),
        "many_pyramids_c4" => many_pyramids(54, 54), // BASELINE config 4: 160,380 cuboids
        "pyramid10" => many_pyramids(1, 1),
        "large_pyramid" => large_pyramid(200),
        "joint_grid" => joint_grid(100),
        "reference_pile" => reference_pile(),
        "large_world" => large_world(1000), // one million static shapes; `large_world [steps]` times the whole run, drops included
        other => panic!("unknown scene {other}"),
    };
    if args.get(2).map(String::as_str) == Some("--dump") {
        let dir = std::path::PathBuf::from(args.get(3).expect("--dump <dir>"));
        std::fs::create_dir_all(&dir).expect("create dump dir");
        let mut done = 0u32;
        for target in [1u32, 10, 100, 120, 1000] {
            while done < target {
                world.step();
                done += 1;
            }
            dump(&world, &dir.join(format!("{scene}_s{target}.rpdump")), target);
        }
        return;
    }
    if args.get(2).map(String::as_str) == Some("--trace") {
        let path = std::path::PathBuf::from(args.get(3).expect("--trace <file> [steps]"));
        let steps: u32 = args.get(4).and_then(|s| s.parse().ok()).unwrap_or(120);
        trace(&mut world, scene, &path, steps);
        return;
    }
    if scene == "large_world" {
        // the benchmark IS the drop phase: no warm-up, a sphere every 5 steps up to 100 (tools/large_world.py: 699 steps after the first)
        let steps: i32 = args.get(2).and_then(|s| s.parse().ok()).unwrap_or(700);
        world.step();
        let (mut dropped, t0) = (0, Instant::now());
        for step in 1..steps {
            if dropped < 100 && step % 5 == 0 {
                large_world_drop(&mut world, dropped, 1000, 100);
                dropped += 1;
            }
            world.step();
        }
        let dt = t0.elapsed().as_secs_f64();
        println!("{{\"scene\": \"large_world\", \"threads\": {}, \"steps\": {}, \"steps_per_s\": {:.2}, \"ms_per_step\": {:.4}}}",
                 rayon::current_num_threads(), steps - 1, (steps - 1) as f64 / dt, dt / (steps - 1) as f64 * 1e3);
        return;
    }
    let warmup: usize = args.get(2).and_then(|s| s.parse().ok()).unwrap_or(60);
    let steps: usize = args.get(3).and_then(|s| s.parse().ok()).unwrap_or(1000);
    for _ in 0..warmup {
        world.step();
    }
    let t0 = Instant::now();
    for _ in 0..steps {
        world.step();
    }
    let dt = t0.elapsed().as_secs_f64();
    println!(
        "{{\"scene\": \"{scene}\", \"threads\": {}, \"steps\": {steps}, \"steps_per_s\": {:.2}, \"ms_per_step\": {:.4}}}",
        rayon::current_num_threads(),
        steps as f64 / dt,
        dt / steps as f64 * 1e3
    );
    println!("{}", world.physics_pipeline.counters);
}
