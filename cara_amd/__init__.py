"""cara_amd -- MI355X-native CaRA fine-tuning hot path (hand-written HIP for gfx950 behind a C ABI)."""
