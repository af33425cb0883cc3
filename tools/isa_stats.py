#!/usr/bin/env python3
"""Dev helper: per-kernel instruction-class counts from a hipcc -S listing (not part of the product)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\S+):\s*;\s*@\S+\n(.*?)\.Lfunc_end\d+:', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name: continue
    c = lambda p: len(re.findall(p, body))
    print(name[:70], '| instr', sum(1 for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')),
          'fma', c(r'\tv_fma_f32|\tv_fmac_f32'), 'mul', c(r'\tv_mul_f32'), 'add', c(r'\tv_add_f32|\tv_sub_f32'),
          'divscale', c('v_div_scale_f32'), 'pk', c(r'\tv_pk_'), 'ldx4', c('global_load_dwordx4'), 'ldx2', c('global_load_dwordx2'),
          'ld', c(r'global_load_'), 'st', c('global_store_'), 'stx4', c('global_store_dwordx4'), 'dsw128', c('ds_write_b128'), 'dsr128', c('ds_read_b128'),
          'sload', c(r'\ts_load'), 'bperm', c('ds_bpermute'), 'dpp', c('_dpp'), 'barrier', c('s_barrier'))
