"""Host-side data utilities of the hot path (random-draw sources, blur-kernel synthesis)."""
