"""Host-side helpers of the VTC step (mirror of the reference's src/core for this path)."""

VTC_LABELS = ["FEM", "MAL", "KCHI", "OCH"]   # reference src/core/__init__.py:7
