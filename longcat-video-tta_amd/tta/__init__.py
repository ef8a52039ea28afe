"""Host side of the test-time-adaptation layer (the part of the hot path the reference owns itself):
latent split, conditioned flow-matching loss, LoRA injection, the forward / LoRA-only backward / fused AdamW
inner loop and anchored early stopping — same names, argument meaning and error behaviour as
delta_experiment/scripts/{common,early_stopping}.py and lora_experiment/scripts/run_lora_tta.py."""
