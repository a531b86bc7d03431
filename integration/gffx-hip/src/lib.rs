//! `gffx-hip`: the Rust side of the drop-in boundary (include/gffx_hip.h).
//!
//! SOURCE ONLY -- never compiled in this repository's build image (no Rust toolchain).  It shows,
//! in the reference's own language, what a GFFx maintainer adds to swap the body of
//! `commands::intersect::query_features` (src/commands/intersect.rs:105-169) and of
//! `commands::depth::compute_hit_depth` (src/commands/depth.rs:222-293) for the MI355X engine.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct gffx_hip_index {
    _p: [u8; 0],
}
#[repr(C)]
pub struct gffx_hip_regions {
    _p: [u8; 0],
}
#[repr(C)]
pub struct gffx_hip_batch {
    _p: [u8; 0],
}
#[repr(C)]
pub struct gffx_hip_depth {
    _p: [u8; 0],
}

pub const GFFX_MODE_CONTAINED: c_int = 0; // OverlapMode::Contained       intersect.rs:75
pub const GFFX_MODE_CONTAINS_REGION: c_int = 1; // OverlapMode::ContainsRegion  intersect.rs:76
pub const GFFX_MODE_OVERLAP: c_int = 2; // OverlapMode::Overlap         intersect.rs:77
pub const GFFX_OUT_FIDS: u32 = 2;
pub const GFFX_OUT_TRIPLES: u32 = 4;
pub const GFFX_OUT_ROOT_BITMAP: u32 = 8;
pub const GFFX_OUT_OFFSETS: u32 = 16;
pub const GFFX_OUT_OFFSETS32: u32 = 64;
pub const GFFX_OUT_SEGBASE: u32 = 256; // one u64 segment base per group of 256 regions (include/gffx_hip.h)
pub const GFFX_OUT_NO_COUNTS: u32 = 512; // with GFFX_OUT_ROOT_BITMAP alone: no per-region counts (the CLI wants the unique roots only)
pub const GFFX_OUT_BITMAP_KEEP: u32 = 128; // with GFFX_OUT_ROOT_BITMAP: accumulate into the bitmap of the passes before (streamed chunks)
pub const GFFX_STRATEGY_AUTO: c_int = 0;

extern "C" {
    pub fn gffx_hip_device_count() -> c_int;
    pub fn gffx_hip_last_error() -> *const c_char;
    pub fn gffx_hip_index_create(
        n_chr: u32,
        chr_offsets: *const u32,
        start: *const u32,
        end: *const u32,
        root_fid: *const u32,
        device: c_int,
        out: *mut *mut gffx_hip_index,
    ) -> c_int;
    pub fn gffx_hip_index_destroy(ix: *mut gffx_hip_index);
    pub fn gffx_hip_index_n_roots(ix: *const gffx_hip_index) -> u64;
    pub fn gffx_hip_index_sorted_fids(ix: *const gffx_hip_index) -> *const u32;
    pub fn gffx_hip_query_features(
        ix: *const gffx_hip_index,
        regions: *const u32,
        nq: u64,
        mode: c_int,
        invert: c_int,
        triples_out: *mut *mut u32,
        n_triples: *mut u64,
    ) -> c_int;
    pub fn gffx_hip_free_host(p: *mut std::ffi::c_void);
    // streaming form
    pub fn gffx_hip_batch_create(ix: *const gffx_hip_index, max_queries: u64, out: *mut *mut gffx_hip_batch) -> c_int;
    pub fn gffx_hip_batch_destroy(b: *mut gffx_hip_batch);
    pub fn gffx_hip_batch_set_regions_host(b: *mut gffx_hip_batch, regions: *const u32, nq: u64) -> c_int;
    pub fn gffx_hip_batch_run(b: *mut gffx_hip_batch, mode: c_int, invert: c_int, out_flags: u32, strategy: c_int) -> c_int;
    pub fn gffx_hip_batch_wait(b: *mut gffx_hip_batch) -> c_int;
    pub fn gffx_hip_batch_sync(b: *mut gffx_hip_batch) -> c_int;
    pub fn gffx_hip_batch_block_threads(b: *const gffx_hip_batch) -> u32;
    pub fn gffx_hip_batch_block_count(b: *const gffx_hip_batch) -> u32;
    pub fn gffx_hip_batch_wide_form(b: *const gffx_hip_batch) -> c_int; // (round 5: the MIXED form -- narrow and wide regions lane by lane)
    // round 5: tuning knobs (read from the environment once per object; changed through set_option), the non-default ones as JSON;
    // the kept pairs of all root passes since the last pass without GFFX_OUT_BITMAP_KEEP (a streaming caller's per-device hit count)
    pub fn gffx_hip_batch_set_option(b: *mut gffx_hip_batch, name: *const std::os::raw::c_char, value: std::os::raw::c_long) -> c_int;
    pub fn gffx_hip_batch_options(b: *const gffx_hip_batch, buf: *mut std::os::raw::c_char, cap: usize) -> c_int;
    pub fn gffx_hip_index_options(ix: *const gffx_hip_index, buf: *mut std::os::raw::c_char, cap: usize) -> c_int;
    pub fn gffx_hip_batch_kept_pairs_accumulated(b: *mut gffx_hip_batch, out: *mut u64) -> c_int;
    // round 6: passes over several batches handed over together -- from four batches on ONE launch serves a group of up to 8 of them
    // (results exactly those of single gffx_hip_batch_run calls); how the passes would be cut into launches
    pub fn gffx_hip_batches_run_n(batches: *const *mut gffx_hip_batch, n_batches: u32, mode: c_int, invert: c_int, out_flags: u32,
                                  strategy: c_int, n_passes: u64) -> c_int;
    pub fn gffx_hip_batches_plan(batches: *const *mut gffx_hip_batch, n_batches: u32, groups: *mut u32, largest: *mut u32, streams: *mut u32) -> c_int;
    // streaming BED ingestion through pinned staging buffers, several GPUs
    pub fn gffx_hip_regions_create(device: c_int, capacity_rows: u64, chunk_rows: u64, keep_all: c_int, out: *mut *mut gffx_hip_regions) -> c_int;
    pub fn gffx_hip_regions_destroy(r: *mut gffx_hip_regions);
    pub fn gffx_hip_regions_staging(r: *mut gffx_hip_regions, k: c_int) -> *mut u32;
    pub fn gffx_hip_regions_wait_staging(r: *mut gffx_hip_regions, k: c_int) -> c_int;
    pub fn gffx_hip_regions_append(r: *mut gffx_hip_regions, k: c_int, n_rows: u64) -> c_int;
    pub fn gffx_hip_batch_set_regions_store(b: *mut gffx_hip_batch, r: *const gffx_hip_regions, k: c_int, first: u64, n_rows: u64) -> c_int;
    pub fn gffx_hip_index_clone(ix: *const gffx_hip_index, device: c_int, out: *mut *mut gffx_hip_index) -> c_int;
    pub fn gffx_hip_allgather_counts(n_dev: c_int, devices: *const c_int, counts_in: *const u64, counts_out: *mut u64) -> c_int;
    pub fn gffx_hip_batch_copy_root_bitmap(b: *mut gffx_hip_batch, host: *mut u64, n_words: u64) -> c_int;
    // depth (BED source)
    pub fn gffx_hip_depth_create(
        device: c_int,
        n_groups: u32,
        n_blocks: u32,
        block_line_off: *const u64,
        line_start: *const u32,
        line_end: *const u32,
        line_group: *const u32,
        n_fid: u32,
        block_of_fid: *const u32,
        out: *mut *mut gffx_hip_depth,
    ) -> c_int;
    pub fn gffx_hip_depth_accumulate(d: *mut gffx_hip_depth, b: *mut gffx_hip_batch) -> c_int;
    pub fn gffx_hip_depth_copy(d: *mut gffx_hip_depth, depth: *mut u64, min_start: *mut u32, max_end: *mut u32) -> c_int;
    pub fn gffx_hip_depth_destroy(d: *mut gffx_hip_depth);
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(gffx_hip_last_error()).to_string_lossy().into_owned() }
}

/// commands/intersect.rs:73-78
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum OverlapMode {
    Contained,
    ContainsRegion,
    Overlap,
}

/// The device-resident form of `TreeIndexData.chr_entries` (src/utils/tree_index.rs:12-16).
pub struct HipIndex(*mut gffx_hip_index);
unsafe impl Send for HipIndex {}
unsafe impl Sync for HipIndex {} // immutable after creation (gffx_hip.h "Threading")
impl Drop for HipIndex {
    fn drop(&mut self) {
        unsafe { gffx_hip_index_destroy(self.0) }
    }
}

impl HipIndex {
    /// `per_seqid[i]` = the (start, end, root_fid) triples of seqid_num i in any order -- what the
    /// builder feeds to `IntervalTree::new` (src/index_builder/core.rs:177-180, :206-218).
    pub fn new(per_seqid: &[Vec<(u32, u32, u32)>], device: i32) -> anyhow::Result<Self> {
        let mut offs = vec![0u32];
        let (mut s, mut e, mut f) = (vec![], vec![], vec![]);
        for v in per_seqid {
            for &(a, b, c) in v {
                s.push(a);
                e.push(b);
                f.push(c);
            }
            offs.push(s.len() as u32);
        }
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            gffx_hip_index_create(per_seqid.len() as u32, offs.as_ptr(), s.as_ptr(), e.as_ptr(), f.as_ptr(), device, &mut h)
        };
        if rc != 0 {
            anyhow::bail!("gffx_hip_index_create: {}", last_error());
        }
        Ok(HipIndex(h))
    }
    pub fn raw(&self) -> *const gffx_hip_index {
        self.0
    }
}

/// One region / one result row as the C-ABI lays them out: three consecutive u32.  A Rust tuple `(u32, u32, u32)` has NO
/// guaranteed layout (`repr(Rust)` may reorder or pad fields), so tuple slices are never cast to `*const u32`: rows are copied
/// into / out of this `repr(C)` type.  (12 bytes per row, once per call: noise next to the PCIe transfer of the same rows.)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct Row3 {
    pub a: u32,
    pub b: u32,
    pub c: u32,
}
const _: () = assert!(std::mem::size_of::<Row3>() == 12 && std::mem::align_of::<Row3>() == 4);

/// Drop-in for `commands::intersect::query_features` (src/commands/intersect.rs:105-169): same
/// arguments, one `(root_fid, iv.start, iv.end)` per kept (region, root) pair, order unspecified
/// (as in the reference, whose order is an FxHashMap walk plus a tree DFS).
pub fn query_features(
    index: &HipIndex,
    regions: &[(u32, u32, u32)],
    mode: OverlapMode,
    invert: bool,
    _verbose: bool,
) -> anyhow::Result<Vec<(u32, u32, u32)>> {
    let (mut p, mut n) = (std::ptr::null_mut::<u32>(), 0u64);
    let m = match mode {
        OverlapMode::Contained => GFFX_MODE_CONTAINED,
        OverlapMode::ContainsRegion => GFFX_MODE_CONTAINS_REGION,
        OverlapMode::Overlap => GFFX_MODE_OVERLAP,
    };
    let rows: Vec<Row3> = regions.iter().map(|&(chr, start, end)| Row3 { a: chr, b: start, c: end }).collect();
    let rc = unsafe {
        gffx_hip_query_features(index.0, rows.as_ptr() as *const u32, rows.len() as u64, m, invert as c_int, &mut p, &mut n)
    };
    if rc != 0 {
        // GFFX_E_CHR_RANGE (-5) stands where the reference panics on `b[chr as usize]` (intersect.rs:117)
        anyhow::bail!("gffx_hip_query_features: {}", last_error());
    }
    let out = unsafe { std::slice::from_raw_parts(p as *const Row3, n as usize) }
        .iter()
        .map(|r| (r.a, r.b, r.c)) // (root_fid, iv.start, iv.end)
        .collect();
    unsafe { gffx_hip_free_host(p as *mut _) };
    Ok(out)
}
