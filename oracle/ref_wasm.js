// TEST INFRASTRUCTURE — not product code.
//
// Driver for the reference's own compiled artefact (pixo v0.4.1 built to
// WebAssembly by the reference's CI, committed upstream at
// web/src/lib/pixo-wasm/pixo_bg.wasm).  `make -C oracle ref` copies that
// BINARY (never sources) to oracle/_ref/pixo_bg.wasm; this file is our own
// loader for it.  It is the strongest oracle available: there is no Rust
// toolchain in the build image, so the reference cannot be compiled natively.
//
// Boundary exercised: `encode_jpeg(data,w,h,color_type,quality,preset,s420)`
// (reference src/wasm.rs:113-142) and `encode_png` (src/wasm.rs:79).
//
// Usage (batch, one process, many cases):
//   node oracle/ref_wasm.js <manifest.json>
// manifest = {"wasm": "...optional path...", "cases": [
//   {"kind":"jpeg","input":"in.bin","w":16,"h":16,"color_type":2,"quality":80,
//    "preset":0,"s420":true,"output":"out.jpg","repeat":1}, ...]}
// For every case prints one JSON line: {"ok":true,"len":N,"ms":[...]} or
// {"ok":false,"error":"<pixo::Error Display string>"}.
'use strict';
const fs = require('fs');
const path = require('path');

function load(wasmPath) {
  const bytes = fs.readFileSync(wasmPath);
  const st = { lastErr: null, wasm: null };
  const imports = { wbg: {
    // the module's only import: constructs a JsError from (ptr,len)
    __wbg_Error_52673b7de5a0ca89: (p, l) => {
      st.lastErr = Buffer.from(st.wasm.memory.buffer, p, l).toString();
      return 132;
    },
  } };
  st.wasm = new WebAssembly.Instance(new WebAssembly.Module(bytes), imports).exports;
  return st;
}

function callEncode(st, fn, data, tailArgs) {
  const wasm = st.wasm;
  const ret = wasm.__wbindgen_add_to_stack_pointer(-16);
  const ptr = wasm.__wbindgen_export(data.length, 1) >>> 0; // malloc(len, align)
  new Uint8Array(wasm.memory.buffer).set(data, ptr);
  st.lastErr = null;
  fn.apply(null, [ret, ptr, data.length].concat(tailArgs));
  const dv = new DataView(wasm.memory.buffer);
  const out = dv.getInt32(ret, true) >>> 0;
  const len = dv.getInt32(ret + 4, true) >>> 0;
  const isErr = dv.getInt32(ret + 12, true);
  wasm.__wbindgen_add_to_stack_pointer(16);
  if (isErr) throw new Error(st.lastErr === null ? 'unknown error' : st.lastErr);
  const res = Buffer.from(new Uint8Array(wasm.memory.buffer).slice(out, out + len));
  wasm.__wbindgen_export2(out, len, 1); // free(ptr, len, align)
  return res;
}

function main() {
  const manifest = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const wasmPath = manifest.wasm || path.join(__dirname, '_ref', 'pixo_bg.wasm');
  const st = load(wasmPath);
  for (const c of manifest.cases) {
    try {
      const data = new Uint8Array(fs.readFileSync(c.input));
      const rep = c.repeat || 1;
      const ms = [];
      let out = null;
      for (let i = 0; i < rep; i++) {
        const t0 = process.hrtime.bigint();
        if (c.kind === 'png') {
          out = callEncode(st, st.wasm.encodePng,
            data, [c.w, c.h, c.color_type, c.preset, c.lossy ? 1 : 0]);
        } else {
          out = callEncode(st, st.wasm.encodeJpeg,
            data, [c.w, c.h, c.color_type, c.quality, c.preset, c.s420 ? 1 : 0]);
        }
        ms.push(Number(process.hrtime.bigint() - t0) / 1e6);
      }
      if (c.output) fs.writeFileSync(c.output, out);
      console.log(JSON.stringify({ ok: true, len: out.length, ms }));
    } catch (e) {
      console.log(JSON.stringify({ ok: false, error: String(e.message) }));
    }
  }
}

main();
