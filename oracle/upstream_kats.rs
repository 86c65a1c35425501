//! oracle/upstream_kats.rs -- the upstream pin kit.  NOT compiled in this repository's image (there is no rustc here): it is the
//! paste-ready test module a maintainer with `cargo` and access to geometryresearch/proof-toolbox drops into the REFERENCE crate to
//! settle, in one `cargo test`, every arkworks / proof-toolbox convention this build restates from memory (oracle/README.md, rows 1-9),
//! and -- last test -- whether upstream's prover reproduces this build's golden proof byte for byte.
//!
//!   cp oracle/upstream_kats.rs  <reference>/barnett-smart-card-protocol/src/discrete_log_cards/upstream_kats.rs
//!   cp tests/golden/{fs_kats,curve_kats,shuffle_stark_m2_n26_s7}.json  <reference>/barnett-smart-card-protocol/src/discrete_log_cards/
//!   # src/discrete_log_cards/mod.rs: add  `#[cfg(test)] mod upstream_kats;`   next to `mod tests;`  [REF mod.rs:27-31]
//!   # Cargo.toml [dev-dependencies]: serde_json = "1", hex = "0.4", rand_chacha = "0.3"
//!   cargo test --release upstream_kats -- --nocapture --test-threads 1
//!
//! Every test prints what upstream produced and what the fixture (this build's oracle, tests/golden/gen_golden.py) holds, then asserts
//! equality.  A row that fails changes proof BYTES only (a new transcript version here), never soundness or the re-encrypted deck.
//! Reviewed against [REF src/discrete_log_cards/mod.rs:37-61,80-84,380-443], [REF examples/parameter_selection.rs:95].
use super::*;
use ark_ec::{AffineCurve, ProjectiveCurve};
use ark_ff::{to_bytes, BigInteger, PrimeField, UniformRand, Zero};
use ark_marlin::rng::FiatShamirRng;
use ark_serialize::CanonicalSerialize;
use ark_std::rand::{RngCore, SeedableRng};
use blake2::Blake2s;
use proof_essentials::utils::permutation::Permutation;
use rand_chacha::ChaCha20Rng;
use serde_json::Value;

type Curve = starknet_curve::Projective;
type Affine = starknet_curve::Affine;
type Fr = starknet_curve::Fr;
type Fq = starknet_curve::Fq;
type Protocol = DLCards<Curve>;

fn fixture(name: &str) -> Value {
    let dir = concat!(env!("CARGO_MANIFEST_DIR"), "/src/discrete_log_cards/");
    serde_json::from_str(&std::fs::read_to_string(format!("{}{}", dir, name)).unwrap()).unwrap()
}
fn unhex(v: &Value) -> Vec<u8> {
    hex::decode(v.as_str().unwrap()).unwrap()
}
fn fr_hex(x: &Fr) -> String {
    // the fixtures print scalars as Python `hex(int)`: big-endian, no leading zeros
    let be = x.into_repr().to_bytes_be();
    let s = hex::encode(be);
    format!("0x{}", s.trim_start_matches('0'))
}
/// wire v1 point = x LE || y LE (32 + 32 bytes on the STARK curve), infinity = all zero
fn wire_point(b: &[u8]) -> Affine {
    if b.iter().all(|v| *v == 0) {
        return Affine::zero();
    }
    let x = Fq::from_le_bytes_mod_order(&b[..32]);
    let y = Fq::from_le_bytes_mod_order(&b[32..64]);
    let p = Affine::new(x, y, false);
    assert!(p.is_on_curve());
    p
}
fn point_wire(p: &Affine) -> Vec<u8> {
    if p.is_zero() {
        return vec![0u8; 64];
    }
    let mut out = p.x.into_repr().to_bytes_le();
    out.extend(p.y.into_repr().to_bytes_le());
    out
}

/// README row 1: FiatShamirRng::<Blake2s>::from_seed / absorb, and row 2: Fr::rand (limb order, shaved bits, Montgomery reading)
#[test]
fn row1_row2_fiat_shamir_rng_and_fr_rand() {
    let k = fixture("fs_kats.json");
    let mut fs = FiatShamirRng::<Blake2s>::from_seed(&to_bytes![SHUFFLE_RNG_SEED].unwrap()); // [REF mod.rs:84,408,436]
    let want = &k["challenges_stark"];
    for i in 0..3 {
        let x = Fr::rand(&mut fs);
        println!("after_seed[{}]  upstream {}  fixture {}", i, fr_hex(&x), want["after_seed"][i]);
        assert_eq!(fr_hex(&x), want["after_seed"][i].as_str().unwrap());
    }
    let bytes: Vec<u8> = (0u8..200).collect();
    fs.absorb(&bytes);
    for i in 0..3 {
        let x = Fr::rand(&mut fs);
        println!("after_absorb[{}] upstream {}  fixture {}", i, fr_hex(&x), want["after_absorb_0_199"][i]);
        assert_eq!(fr_hex(&x), want["after_absorb_0_199"][i].as_str().unwrap());
    }
    // the ChaCha20 word order under it (rand_chacha, zero key): first 64 bytes of next_u64 output, little-endian
    let mut z = ChaCha20Rng::from_seed([0u8; 32]);
    let mut first = Vec::new();
    for _ in 0..8 {
        first.extend_from_slice(&z.next_u64().to_le_bytes());
    }
    assert_eq!(hex::encode(first), k["chacha20_zero_key_first64"].as_str().unwrap());
}

/// README row 3: to_bytes! of an affine point = x LE || y LE || 1 flag byte (65 bytes here), slices without a length prefix
#[test]
fn row3_to_bytes_layout() {
    let g = Affine::prime_subgroup_generator();
    let b = to_bytes![g].unwrap();
    println!("to_bytes![G].len() = {} (this build absorbs 65)", b.len());
    assert_eq!(b.len(), 65);
    assert_eq!(&b[..64], &point_wire(&g)[..]);
    assert_eq!(b[64], 0);
    assert_eq!(to_bytes![vec![g, g]].unwrap().len(), 130);
}

/// README row 7: Permutation::permute_array(v)[i] = v[mapping[i]]
#[test]
fn row7_permute_array() {
    let p = Permutation::from(&vec![2usize, 0, 1]);
    let v = vec!['a', 'b', 'c'];
    let out = p.permute_array(&v);
    println!("permute_array([a,b,c]) under [2,0,1] = {:?} (this build: [c, a, b])", out);
    assert_eq!(out, vec!['c', 'a', 'b']);
}

/// README row 8: C::rand ("setup v2"): x = Fq::rand, greatest = rng.gen::<bool>(), get_point_from_x, cofactor cleared
#[test]
fn row8_curve_rand() {
    let g = fixture("shuffle_stark_m2_n26_s7.json");
    // gen_golden.py derives the parameters of this fixture from ChaCha20Rng::from_seed(Blake2s("setup" || seed)): the first point it
    // draws is the ElGamal generator, params[0..64]
    let params = unhex(&g["params"]);
    let seed = unhex(&g["setup_seed"]);
    let mut key = [0u8; 32];
    key.copy_from_slice(&seed);
    let p = Curve::rand(&mut ChaCha20Rng::from_seed(key)).into_affine();
    println!("C::rand  upstream {}  fixture {}", hex::encode(point_wire(&p)), hex::encode(&params[..64]));
    assert_eq!(point_wire(&p), params[..64].to_vec());
}

/// README row 9: compressed CanonicalSerialize of a point (SURVEY.md App. B known answer) and of a scalar
#[test]
fn row9_canonical_point() {
    let k = fixture("curve_kats.json");
    let kg = wire_point(&unhex(&k["stark"]["kG"]));
    let mut out = Vec::new();
    kg.serialize(&mut out).unwrap();
    println!("serialize(kG) = {}", hex::encode(&out));
    assert_eq!(hex::encode(&out), "c1776893d3f3f9ce89a577f5b003bac8d9de258967f8ef2dfb23c41171346a05");
    assert_eq!(out.len(), 32);
}

/// the re-encryption itself (pinned mathematically): Remask on the fixture's card under the fixture's key
#[test]
fn remask_known_answer() {
    let k = fixture("curve_kats.json");
    let r = &k["stark"]["remask"];
    let pk = wire_point(&unhex(&r["pk"]));
    let ct = unhex(&r["ct"]);
    let card = MaskedCard::<Curve>(wire_point(&ct[..64]), wire_point(&ct[64..]));
    let alpha = Fr::from_le_bytes_mod_order(&unhex(&r["alpha"]));
    let pp = el_gamal::Parameters::<Curve> { generator: Affine::prime_subgroup_generator() };
    let out = card.remask(&pp, &pk, &alpha).unwrap(); // [REF remasking.rs:9-22]
    let mut got = point_wire(&out.0);
    got.extend(point_wire(&out.1));
    assert_eq!(hex::encode(got), r["out"].as_str().unwrap());
}

/// README rows 4, 5, 6 in one: upstream's prover on this build's golden inputs, prover randomness from ChaCha20Rng::from_seed(seed).
/// Equal bytes => transcript order, draw order and proof contents all agree and "parity unpinned" is lifted.  Unequal: the printed
/// element-by-element comparison (wire order of DESIGN.md section 2) says where the two diverge first.
#[test]
fn rows4_5_6_golden_proof() {
    let g = fixture("shuffle_stark_m2_n26_s7.json");
    let (m, n) = (g["m"].as_u64().unwrap() as usize, g["n"].as_u64().unwrap() as usize);
    let raw = unhex(&g["params"]); // G | ck_0..ck_{n-1} | H | gen
    let pt = |i: usize| wire_point(&raw[64 * i..64 * (i + 1)]);
    let enc = el_gamal::Parameters::<Curve> { generator: pt(0) };
    let ck = pedersen::CommitKey::<Curve>::new((1..=n).map(pt).collect(), pt(n + 1));
    let pp = Parameters::<Curve>::new(m, n, enc, ck, pt(n + 2)); // [REF mod.rs:37-61]
    let pk = wire_point(&unhex(&g["pk"]));
    let deck_b = unhex(&g["deck"]);
    let deck: Vec<MaskedCard<Curve>> =
        deck_b.chunks(128).map(|c| el_gamal::Ciphertext(wire_point(&c[..64]), wire_point(&c[64..]))).collect();
    let rho: Vec<Fr> = unhex(&g["rho"]).chunks(32).map(Fr::from_le_bytes_mod_order).collect();
    let perm: Vec<usize> = g["perm"].as_array().unwrap().iter().map(|v| v.as_u64().unwrap() as usize).collect();
    let mut key = [0u8; 32];
    key.copy_from_slice(&unhex(&g["prover_seed"]));
    let mut rng = ChaCha20Rng::from_seed(key);
    let (shuffled, proof) =
        Protocol::shuffle_and_remask(&mut rng, &pp, &pk, &deck, &rho, &Permutation::from(&perm)).unwrap(); // [REF mod.rs:380-418]
    let mut got = Vec::new();
    for c in &shuffled {
        got.extend(point_wire(&c.0));
        got.extend(point_wire(&c.1));
    }
    assert_eq!(hex::encode(&got), g["shuffled"].as_str().unwrap(), "re-encrypted deck (must agree: SURVEY.md 8c4)");
    assert!(Protocol::verify_shuffle(&pp, &pk, &deck, &shuffled, &proof).is_ok()); // [REF mod.rs:420-443]
    println!("proof.serialized_size() = {} (this build's grouping at (2,26): see mp_serialized_proof_size)", proof.serialized_size());
    // wire v1 of upstream's proof: every group element uncompressed x || y, every scalar 32 B LE, in the order of DESIGN.md section 2.
    // The struct lives in proof-toolbox (shuffle::proof::Proof); write the projection once upstream's field names are in front of
    // you, e.g.  for c in &proof.a_commits { wire.extend(point_wire(c)) } ... and compare with g["proof"].
    let mut bytes = Vec::new();
    proof.serialize_uncompressed(&mut bytes).unwrap();
    println!("upstream proof, serialize_uncompressed: {} bytes\n{}", bytes.len(), hex::encode(&bytes));
    println!("this build's wire v1 proof: {} bytes\n{}", g["proof"].as_str().unwrap().len() / 2, g["proof"].as_str().unwrap());
}
