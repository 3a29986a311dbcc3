"""`gsplat.cuda` namespace (module path kept for the reference's import at basics.py:15)."""
