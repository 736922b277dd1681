cd tools/ubench
for m in 0 32 64 96 4 68 100; do echo "=== SK_SKIP=$m"; SK_SKIP=$m ./sk_base p | grep -E -A1 "main loop" | grep -v "^--" | cut -c1-200 | head -16; done
