"""Model wrappers around the gated-token backbone (counterparts of the reference's models/ directory)."""
