"""Components written ahead of their hardware validation. Nothing here is imported by the package's
public surface (`pytorch_generative_amd.nn`, `.models`, `compat`), by bench.py or by the default test
tiers; each module states what is still unverified."""
