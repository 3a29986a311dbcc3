"""`gsplat.cuda_legacy` namespace (module path kept for the reference's imports at basics.py:13-14)."""
