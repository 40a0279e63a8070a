//! Criterion bench of the forward transform, N = 2^11 .. 2^16, input a[i] = i -- the measurement the
//! crate's benches/ntt.rs sets out to make.  That file imports a crate called `app` (benches/ntt.rs:2-3)
//! while the package is `ntt` (Cargo.toml:2), so it does not compile, and it builds ONE table of 2^16
//! points and feeds it shorter slices (benches/ntt.rs:7-8), which only the forward direction of the CPU
//! loops tolerates.  This one names the crate correctly and builds a table per size, with nested roots
//! psi_n = psi_{2^16}^(2^16 / n), so the same file measures the CPU crate and the GPU-backed one.
//!
//! Not compiled in this repository's build image (no Rust toolchain); bench.py is the measured
//! benchmark of the GPU path.
use criterion::{criterion_group, criterion_main, BenchmarkId, Criterion};
use ntt::dft::ntt::Table;
use ntt::dft::DFT;
use ntt::utils::mod_exp_u64;

const Q: u64 = 0x1fffffffffe00001;
const PSI_2_16: u64 = 0x15eb043c7aa2b01f; // the root of Table::new(): a primitive 2^17-th root of unity

fn forward_inplace(c: &mut Criterion) {
    let mut group = c.benchmark_group("forward_inplace");
    for log_n in 11..17u32 {
        let n: u64 = 1 << log_n;
        // even log2 n only: the crate's bit_reverse (utils.rs:14-17) is wrong for odd bit counts
        if log_n % 2 != 0 {
            continue;
        }
        let table = Table::<u64>::from_prime_and_root(Q, mod_exp_u64(PSI_2_16, (1 << 16) / n, Q), n);
        let mut a: Vec<u64> = (0..n).collect();
        group.bench_with_input(BenchmarkId::new("prime", n), &(), |b, _| {
            b.iter(|| table.forward_inplace(&mut a))
        });
    }
    group.finish();
}

fn forward_backward_pairs(c: &mut Criterion) {
    // the README loop (src/main.rs:26-31): forward then backward on the same buffer
    let table = Table::<u64>::new();
    let mut a: Vec<u64> = (0..1u64 << 16).collect();
    c.bench_function("fwd_inv_pair/65536", |b| {
        b.iter(|| {
            table.forward_inplace(&mut a);
            table.backward_inplace(&mut a);
        })
    });
}

criterion_group!(benches, forward_inplace, forward_backward_pairs);
criterion_main!(benches);
