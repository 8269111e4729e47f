"""Action / reward constants (mirror of tetris_gymnasium/mappings)."""
