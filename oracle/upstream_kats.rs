//! oracle/upstream_kats.rs -- the upstream pin kit.  NOT compiled here (no rustc in this image): a maintainer with `cargo` and access to
//! geometryresearch/proof-toolbox drops it into the REFERENCE crate to settle, in one run, every arkworks / proof-toolbox convention this
//! build restates from memory (oracle/README.md rows 1-9) and whether upstream's prover reproduces this build's golden proof.
//!   cp oracle/upstream_kats.rs <ref>/barnett-smart-card-protocol/src/discrete_log_cards/upstream_kats.rs
//!   cp tests/golden/{fs_kats,curve_kats,shuffle_stark_m2_n26_s7,shuffle_stark_m4_n13_s9,chain_stark_m2_n3_L3_s21}.json <ref>/barnett-smart-card-protocol/src/discrete_log_cards/
//!   mod.rs: add `#[cfg(test)] mod upstream_kats;` next to `mod tests;` [REF mod.rs:27-31]
//!   Cargo.toml [dev-dependencies]: serde_json = "1", hex = "0.4", rand_chacha = "0.3"
//!   cargo test --release upstream_kats -- --nocapture --test-threads 1
//! A failing row changes proof BYTES only (a new transcript version here), never soundness or the re-encrypted deck.
//! Reviewed against [REF src/discrete_log_cards/mod.rs:37-61,80-84,380-443] and [REF examples/parameter_selection.rs:95].
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

fn fixture(name: &str) -> Value {
    let p = format!("{}/src/discrete_log_cards/{}", env!("CARGO_MANIFEST_DIR"), name);
    serde_json::from_str(&std::fs::read_to_string(p).unwrap()).unwrap()
}
fn unhex(v: &Value) -> Vec<u8> { hex::decode(v.as_str().unwrap()).unwrap() }
/// fixtures print scalars as Python `hex(int)`
fn fr_hex(x: &Fr) -> String { format!("0x{}", hex::encode(x.into_repr().to_bytes_be()).trim_start_matches('0')) }
/// wire v1 point = x LE || y LE (2 x 32 bytes), infinity = all zero
fn wire_point(b: &[u8]) -> Affine {
    if b.iter().all(|v| *v == 0) { return Affine::zero(); }
    let p = Affine::new(Fq::from_le_bytes_mod_order(&b[..32]), Fq::from_le_bytes_mod_order(&b[32..64]), false);
    assert!(p.is_on_curve());
    p
}
fn point_wire(p: &Affine) -> Vec<u8> {
    if p.is_zero() { return vec![0u8; 64]; }
    [p.x.into_repr().to_bytes_le(), p.y.into_repr().to_bytes_le()].concat()
}
fn seed32(b: &[u8]) -> [u8; 32] { let mut k = [0u8; 32]; k[..b.len()].copy_from_slice(b); k }

/// rows 1, 2: FiatShamirRng::<Blake2s>::from_seed / absorb; Fr::rand (limb order, shaved bits, Montgomery reading); ChaCha20 word order
#[test]
fn row1_row2_fiat_shamir_rng_and_fr_rand() {
    let k = fixture("fs_kats.json");
    let mut fs = FiatShamirRng::<Blake2s>::from_seed(&to_bytes![SHUFFLE_RNG_SEED].unwrap()); // [REF mod.rs:84,408,436]
    for (key, absorb) in [("after_seed", false), ("after_absorb_0_199", true)] {
        if absorb { fs.absorb(&(0u8..200).collect::<Vec<u8>>()); }
        for i in 0..3 {
            let x = Fr::rand(&mut fs);
            println!("{}[{}] upstream {} fixture {}", key, i, fr_hex(&x), k["challenges_stark"][key][i]);
            assert_eq!(fr_hex(&x), k["challenges_stark"][key][i].as_str().unwrap());
        }
    }
    let mut z = ChaCha20Rng::from_seed([0u8; 32]);
    let first: Vec<u8> = (0..8).flat_map(|_| z.next_u64().to_le_bytes()).collect();
    assert_eq!(hex::encode(first), k["chacha20_zero_key_first64"].as_str().unwrap());
}

/// row 3: to_bytes! of an affine point = x LE || y LE || 1 flag byte; slices without a length prefix
#[test]
fn row3_to_bytes_layout() {
    let g = Affine::prime_subgroup_generator();
    let b = to_bytes![g].unwrap();
    assert_eq!((b.len(), &b[..64], b[64]), (65, &point_wire(&g)[..], 0));
    assert_eq!(to_bytes![vec![g, g]].unwrap().len(), 130);
}

/// row 7: Permutation::permute_array(v)[i] = v[mapping[i]]
#[test]
fn row7_permute_array() {
    assert_eq!(Permutation::from(&vec![2usize, 0, 1]).permute_array(&vec!['a', 'b', 'c']), vec!['c', 'a', 'b']);
}

/// row 8: C::rand ("setup v2"): x = Fq::rand, greatest = rng.gen::<bool>(), get_point_from_x, cofactor.  The fixture's parameters are
/// drawn from ChaCha20Rng::from_seed(seed as u64 LE, zero-padded) (oracle/py/mp_oracle.py gen_inputs): the first point is G = params[0..64]
#[test]
fn row8_curve_rand() {
    let g = fixture("shuffle_stark_m2_n26_s7.json");
    let mut rng = ChaCha20Rng::from_seed(seed32(&g["seed"].as_u64().unwrap().to_le_bytes()));
    let p = Curve::rand(&mut rng).into_affine();
    println!("C::rand upstream {} fixture {}", hex::encode(point_wire(&p)), &g["params"].as_str().unwrap()[..128]);
    assert_eq!(point_wire(&p), unhex(&g["params"])[..64].to_vec());
}

/// row 9: compressed CanonicalSerialize of a point (SURVEY.md App. B known answer)
#[test]
fn row9_canonical_point() {
    let mut out = Vec::new();
    wire_point(&unhex(&fixture("curve_kats.json")["stark"]["kG"])).serialize(&mut out).unwrap();
    assert_eq!(hex::encode(&out), "c1776893d3f3f9ce89a577f5b003bac8d9de258967f8ef2dfb23c41171346a05");
}

/// the re-encryption itself (pinned mathematically) [REF remasking.rs:9-22]
#[test]
fn remask_known_answer() {
    let r = &fixture("curve_kats.json")["stark"]["remask"];
    let ct = unhex(&r["ct"]);
    let card = MaskedCard::<Curve>(wire_point(&ct[..64]), wire_point(&ct[64..]));
    let pp = el_gamal::Parameters::<Curve> { generator: Affine::prime_subgroup_generator() };
    let out = card.remask(&pp, &wire_point(&unhex(&r["pk"])), &Fr::from_le_bytes_mod_order(&unhex(&r["alpha"]))).unwrap();
    assert_eq!(hex::encode([point_wire(&out.0), point_wire(&out.1)].concat()), r["out"].as_str().unwrap());
}

fn table_params(raw: &[u8], m: usize, n: usize) -> Parameters<Curve> {
    let pt = |i: usize| wire_point(&raw[64 * i..64 * (i + 1)]); // G | ck_0..ck_{n-1} | H | gen
    let ck = pedersen::CommitKey::<Curve>::new((1..=n).map(pt).collect(), pt(n + 1));
    Parameters::<Curve>::new(m, n, el_gamal::Parameters { generator: pt(0) }, ck, pt(n + 2)) // [REF mod.rs:37-61]
}
fn wire_deck(b: &[u8]) -> Vec<MaskedCard<Curve>> {
    b.chunks(128).map(|c| el_gamal::Ciphertext(wire_point(&c[..64]), wire_point(&c[64..]))).collect()
}
fn deck_wire(d: &[MaskedCard<Curve>]) -> String {
    hex::encode(d.iter().flat_map(|c| [point_wire(&c.0), point_wire(&c.1)].concat()).collect::<Vec<u8>>())
}
/// one shuffle_and_remask + verify_shuffle of upstream on fixture inputs [REF mod.rs:380-443]; prover randomness from
/// ChaCha20Rng::from_seed(prover_seed).  Returns the re-encrypted deck (wire v1, hex) and upstream's proof
fn upstream_link(pp: &Parameters<Curve>, pk: &Affine, deck: &[MaskedCard<Curve>], link: &Value) -> (String, ZKProofShuffle<Curve>) {
    let rho: Vec<Fr> = unhex(&link["rho"]).chunks(32).map(Fr::from_le_bytes_mod_order).collect();
    let perm: Vec<usize> = link["perm"].as_array().unwrap().iter().map(|v| v.as_u64().unwrap() as usize).collect();
    let mut rng = ChaCha20Rng::from_seed(seed32(&unhex(&link["prover_seed"])));
    let (shuffled, proof) = DLCards::<Curve>::shuffle_and_remask(&mut rng, pp, pk, &deck.to_vec(), &rho, &Permutation::from(&perm)).unwrap();
    assert!(DLCards::<Curve>::verify_shuffle(pp, pk, &deck.to_vec(), &shuffled, &proof).is_ok());
    (deck_wire(&shuffled), proof)
}

/// rows 4, 5, 6: upstream's prover on this build's golden inputs -- the headline shape (2, 26) and the reference's OWN test shape
/// (4, 13) [REF tests.rs:178-179].  Equal bytes => transcript order, draw order and proof contents agree and "parity unpinned" is
/// lifted; unequal: both are printed
#[test]
fn rows4_5_6_golden_proof() {
    for name in ["shuffle_stark_m2_n26_s7.json", "shuffle_stark_m4_n13_s9.json"] {
        let g = fixture(name);
        let (m, n) = (g["m"].as_u64().unwrap() as usize, g["n"].as_u64().unwrap() as usize);
        let pp = table_params(&unhex(&g["params"]), m, n);
        let pk = wire_point(&unhex(&g["pk"]));
        let (got, proof) = upstream_link(&pp, &pk, &wire_deck(&unhex(&g["deck"])), &g);
        assert_eq!(got, g["shuffled"].as_str().unwrap(), "{}: re-encrypted deck (must agree: SURVEY.md 8c4)", name);
        // row 6: this build's grouping gives mp_serialized_proof_size(STARK, m, n); upstream's struct decides the real figure
        println!("{}: proof.serialized_size() = {}", name, proof.serialized_size()); // [REF examples/parameter_selection.rs:95]
        let mut bytes = Vec::new();
        proof.serialize_uncompressed(&mut bytes).unwrap();
        // wire v1 = every group element x || y uncompressed, every scalar 32 B LE, in the order of DESIGN.md section 2: project
        // upstream's struct fields into that order once they are in front of you and compare with g["proof"]
        println!("upstream proof (serialize_uncompressed, {} B): {}", bytes.len(), hex::encode(&bytes));
        println!("this build's wire v1 proof ({} B): {}", g["proof"].as_str().unwrap().len() / 2, g["proof"].as_str().unwrap());
    }
}

/// the chain fixture: one table, three dependent shuffles under one aggregate key, deck j + 1 = output of link j
/// [REF examples/round.rs:268-350]; every deck of the chain must agree (the decks do not depend on the transcript)
#[test]
fn chain_fixture() {
    let g = fixture("chain_stark_m2_n3_L3_s21.json");
    let (m, n) = (g["m"].as_u64().unwrap() as usize, g["n"].as_u64().unwrap() as usize);
    let pp = table_params(&unhex(&g["params"]), m, n);
    let pk = wire_point(&unhex(&g["pk"]));
    let mut deck = wire_deck(&unhex(&g["decks"][0]));
    for (j, link) in g["chain"].as_array().unwrap().iter().enumerate() {
        let (got, proof) = upstream_link(&pp, &pk, &deck, link);
        assert_eq!(got, g["decks"][j + 1].as_str().unwrap(), "deck {} of the chain", j + 1);
        let mut bytes = Vec::new();
        proof.serialize_uncompressed(&mut bytes).unwrap();
        println!("link {}: upstream {} | wire v1 {}", j, hex::encode(&bytes), link["proof"].as_str().unwrap());
        deck = wire_deck(&hex::decode(&got).unwrap());
    }
}
