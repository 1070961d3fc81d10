//! The reference's three PSF benchmarks (benches/psf.rs:26-100) on the GPU types, same names suffixed " (MI355X)", plus the batched
//! form the library is built for.  One `samp_p` call is one unit of work; the trapdoor is generated outside the timed closure and the
//! target is uniform, exactly as in the reference (benches/psf.rs:35-38, :60-65, :87-92).  UNVERIFIED SOURCE (see ../Cargo.toml).
use criterion::*;
use qfall_math::{integer_mod_q::MatZq, rational::Q};
use qfall_tools::primitive::psf::{PSF, PSFGPV, PSFPerturbation};
use qfall_tools::sample::g_trapdoor::gadget_parameters::GadgetParameters;
use qfall_tools_mi355x::{GpuPSFGPV, GpuPSFPerturbation};

/// benches/psf.rs:26-39
fn bench_psf(c: &mut Criterion) {
    let (n, q) = (8, 128);
    let psf = GpuPSFGPV::new(PSFGPV { gp: GadgetParameters::init_default(n, q), s: Q::from(30) * Q::from(n).log(2).unwrap() }, 0, 1);
    let target = MatZq::sample_uniform(n, 1, q);
    let (a, r) = psf.trap_gen();
    c.bench_function("PSF GPV n=8 (MI355X)", |b| b.iter(|| psf.samp_p(&a, &r, &target)));
}

/// benches/psf.rs:51-66
fn bench_psf_perturbation(c: &mut Criterion) {
    let (n, q) = (8, 128);
    let psf = GpuPSFPerturbation::new(PSFPerturbation { gp: GadgetParameters::init_default(n, q), s: Q::from(30), r: Q::from(n).log(2).unwrap() }, 0, 2);
    let target = MatZq::sample_uniform(n, 1, q);
    let (a, r) = psf.trap_gen();
    c.bench_function("PSF Perturbation n=8 (MI355X)", |b| b.iter(|| psf.samp_p(&a, &r, &target)));
}

/// benches/psf.rs:78-93
fn bench_psf_perturbation_larger(c: &mut Criterion) {
    let (n, q) = (64, 128);
    let psf = GpuPSFPerturbation::new(PSFPerturbation { gp: GadgetParameters::init_default(n, q), s: Q::from(100), r: Q::from(n).log(2).unwrap() }, 0, 3);
    let target = MatZq::sample_uniform(n, 1, q);
    let (a, r) = psf.trap_gen();
    c.bench_function("PSF Perturbation n=64 (MI355X)", |b| b.iter(|| psf.samp_p(&a, &r, &target)));
}

/// The same parameter set as above, 4096 independent preimages per call: what the device path is designed for.
fn bench_psf_perturbation_batched(c: &mut Criterion) {
    let (n, q, batch) = (64, 128, 4096);
    let psf = GpuPSFPerturbation::new(PSFPerturbation { gp: GadgetParameters::init_default(n, q), s: Q::from(100), r: Q::from(n).log(2).unwrap() }, 0, 4);
    let targets = MatZq::sample_uniform(batch, n, q);
    let (a, r) = psf.trap_gen();
    let mut group = c.benchmark_group("PSF Perturbation n=64 batch=4096 (MI355X)");
    group.throughput(Throughput::Elements(batch as u64));
    group.bench_function("samp_p_batch", |b| b.iter(|| psf.samp_p_batch(&a, &r, &targets)));
    group.finish();
}

criterion_group!(benches, bench_psf, bench_psf_perturbation, bench_psf_perturbation_larger, bench_psf_perturbation_batched,);
