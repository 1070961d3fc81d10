// criterion entry point, as benches/benchmarks.rs:10-14 of the reference
use criterion::criterion_main;

pub mod psf;

criterion_main! {psf::benches}
