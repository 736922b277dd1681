#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 PMC counters from <prefix>_counter_collection.csv."""
import sys, pandas as pd
df = pd.read_csv(sys.argv[1])
if len(sys.argv) > 3:          # only kernels whose name contains this
    df = df[df['Kernel_Name'].str.contains(sys.argv[3], regex=False)]
df['k'] = df['Kernel_Name'].str.slice(0, 48)
t = df.pivot_table(index='k', columns='Counter_Name', values='Counter_Value', aggfunc='mean')
n = df.groupby('k')['Dispatch_Id'].nunique().rename('dispatches')
pd.set_option('display.width', 250); pd.set_option('display.max_columns', 30); pd.set_option('display.float_format', lambda x: '%.4g' % x)
print(t.join(n).sort_values(t.columns[0], ascending=False).head(int(sys.argv[2]) if len(sys.argv) > 2 else 12))
