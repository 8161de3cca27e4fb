//! Start-up self-test of the memory layouts the C ABI relies on.
//!
//! `bls12_381`'s `Scalar`, `G1Affine` and `G2Affine` are not `repr(C)`; the library reads them as raw bytes
//! (`bh_bases_register(stride, inf_offset)`, Montgomery scalars).  Instead of trusting a compiler's field
//! order, the layout is PROBED once with values whose encodings are known, and the device path is only
//! enabled when every probe agrees.  Otherwise `Layout::probe()` returns `None` and bellman keeps its CPU
//! path (the shim never guesses).
//!
//! What is probed (include/bellman_hip.h "DATA FORMATS"):
//!   * `Scalar`   : 32 bytes, 4 x u64 little-endian Montgomery limbs (R = 2^256): `Scalar::one()` must read
//!                  as R mod q, and `to_bytes()` of a random value must equal the Montgomery reduction of
//!                  its raw limbs.
//!   * `G1Affine` : x at some offset, y at some offset (48 bytes each, Montgomery, R = 2^384 mod p) and one
//!                  `infinity` flag byte; the generator's coordinates are located by searching for the
//!                  byte strings of `Fp::one()`-scaled known values, the flag by comparing the identity with
//!                  a non-identity point.
//!   * `G2Affine` : x.c0 | x.c1 | y.c0 | y.c1 likewise.
use bls12_381::{G1Affine, G2Affine, Scalar};
use std::mem::size_of;

/// Where the library finds the coordinates and the identity flag inside a Rust affine point.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct PointLayout {
    /// size_of::<Affine>() - the stride handed to `bh_bases_register`
    pub stride: usize,
    /// byte offset of the `infinity: Choice` flag
    pub inf_offset: usize,
    /// true when the coordinate block starts at offset 0 in the library's order (x | y, resp.
    /// x.c0 | x.c1 | y.c0 | y.c1); the only arrangement `bh_bases_register` accepts without repacking
    pub coords_at_zero: bool,
}

#[derive(Clone, Copy, Debug)]
pub struct Layout {
    pub g1: PointLayout,
    pub g2: PointLayout,
}

/// Montgomery limbs (little-endian bytes) of 1 in Fr: R = 2^256 mod q
const FR_ONE_MONT: [u64; 4] = [0x0000_0001_ffff_fffe, 0x5884_b7fa_0003_4802, 0x998c_4fef_ecbc_4ff5, 0x1824_b159_acc5_056f];

/// Montgomery limbs of the G1 generator (x | y), R = 2^384 mod p - the same constants bench.py uploads
const G1_GEN_MONT: [u64; 12] = [
    0x5cb3_8790_fd53_0c16, 0x7817_fc67_9976_fff5, 0x154f_95c7_143b_a1c1, 0xf0ae_6acd_f3d0_e747, 0xedce_6ecc_21db_f440,
    0x1201_7741_9e0b_fb75, 0xbaac_93d5_0ce7_2271, 0x8c22_631a_7918_fd8e, 0xdd59_5f13_5707_25ce, 0x51ac_5829_5040_5194,
    0x0e1c_8c3f_ad00_59c0, 0x0bbc_3efc_5008_a26a,
];

/// Montgomery limbs of the G2 generator (x.c0 | x.c1 | y.c0 | y.c1)
const G2_GEN_MONT: [u64; 24] = [
    0xf5f2_8fa2_0294_0a10, 0xb3f5_fb26_87b4_961a, 0xa1a8_93b5_3e2a_e580, 0x9894_999d_1a3c_aee9, 0x6f67_b763_1863_366b,
    0x0581_9192_4350_bcd7, 0xa5a9_c075_9e23_f606, 0xaaa0_c59d_bccd_60c3, 0x3bb1_7e18_e286_7806, 0x1b1a_b6cc_8541_b367,
    0xc2b6_ed0e_f215_8547, 0x1192_2a09_7360_edf3, 0x4c73_0af8_6049_4c4a, 0x597c_fa1f_5e36_9c5a, 0xe7e6_856c_aa0a_635a,
    0xbbef_b5e9_6e0d_495f, 0x07d3_a975_f0ef_25a2, 0x0083_fd8e_7e80_dae5, 0xadc0_fc92_df64_b05d, 0x18aa_270a_2b14_61dc,
    0x86ad_ac6a_3be4_eba0, 0x7949_5c4e_c93d_a33a, 0xe717_5850_a43c_caed, 0x0b2b_c2a1_63de_1bf2,
];

fn bytes_of<T>(v: &T) -> &[u8] {
    // reading the object representation of a plain-old-data value
    unsafe { std::slice::from_raw_parts(v as *const T as *const u8, size_of::<T>()) }
}

fn limbs_to_bytes(limbs: &[u64]) -> Vec<u8> {
    limbs.iter().flat_map(|l| l.to_le_bytes()).collect()
}

fn find(haystack: &[u8], needle: &[u8]) -> Option<usize> {
    haystack.windows(needle.len()).position(|w| w == needle)
}

fn probe_point<T>(generator: &T, identity: &T, gen_coords: &[u64]) -> Option<PointLayout> {
    let g = bytes_of(generator);
    let o = bytes_of(identity);
    let coords = limbs_to_bytes(gen_coords);
    let at = find(g, &coords)?;
    // the flag: exactly one byte outside the coordinate block that is 0 for the generator and non-zero
    // for the identity (padding bytes may hold anything, so they are not required to be equal)
    let mut flag = None;
    for i in 0..g.len() {
        if i >= at && i < at + coords.len() {
            continue;
        }
        if g[i] == 0 && o[i] == 1 {
            if flag.is_some() {
                return None; // ambiguous
            }
            flag = Some(i);
        }
    }
    Some(PointLayout { stride: size_of::<T>(), inf_offset: flag?, coords_at_zero: at == 0 })
}

impl Layout {
    /// `None` = something is not as the C ABI expects: do not use the device path.
    pub fn probe() -> Option<Layout> {
        if size_of::<Scalar>() != 32 {
            return None;
        }
        if bytes_of(&Scalar::one()) != limbs_to_bytes(&FR_ONE_MONT).as_slice() {
            return None;
        }
        // 2 in Montgomery form is 2R mod q: check against the library-independent definition
        let two = Scalar::one() + Scalar::one();
        let mut r2 = [0u64; 4];
        let mut carry = 0u128;
        for i in 0..4 {
            let t = (FR_ONE_MONT[i] as u128) * 2 + carry;
            r2[i] = t as u64;
            carry = t >> 64;
        }
        // 2R < 2q holds (R < q), one conditional subtraction of q suffices
        const Q: [u64; 4] = [0xffff_ffff_0000_0001, 0x53bd_a402_fffe_5bfe, 0x3339_d808_09a1_d805, 0x73ed_a753_299d_7d48];
        let ge = (0..4).rev().find_map(|i| if r2[i] != Q[i] { Some(r2[i] > Q[i]) } else { None }).unwrap_or(true);
        if ge {
            let mut borrow = 0i128;
            for i in 0..4 {
                let t = r2[i] as i128 - Q[i] as i128 - borrow;
                r2[i] = t as u64;
                borrow = if t < 0 { 1 } else { 0 };
            }
        }
        if bytes_of(&two) != limbs_to_bytes(&r2).as_slice() {
            return None;
        }
        let g1 = probe_point(&G1Affine::generator(), &G1Affine::identity(), &G1_GEN_MONT)?;
        let g2 = probe_point(&G2Affine::generator(), &G2Affine::identity(), &G2_GEN_MONT)?;
        if !g1.coords_at_zero || !g2.coords_at_zero {
            return None; // bh_bases_register wants the coordinates first; repacking is the caller's fallback
        }
        Some(Layout { g1, g2 })
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn layout_is_what_the_c_abi_documents() {
        // with bls12_381 0.8.0 on x86-64 this is stride 104 / 200 and the flag right after the coordinates
        let l = Layout::probe().expect("layout probe failed: the device path would be disabled");
        assert_eq!(l.g1, PointLayout { stride: 104, inf_offset: 96, coords_at_zero: true });
        assert_eq!(l.g2, PointLayout { stride: 200, inf_offset: 192, coords_at_zero: true });
    }
}
