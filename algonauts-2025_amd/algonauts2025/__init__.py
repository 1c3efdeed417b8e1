"""algonauts2025 (MI355X build): `model.FmriEncoder(Config)`, `pl_module.BrainModule` and the
`main` entry points of the reference package, with the encode hot path on gfx950 HIP kernels."""
