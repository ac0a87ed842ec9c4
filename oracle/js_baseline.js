'use strict';
/* Timed "JavaScript without a GPU" baseline (BASELINE.md §4, CPU-JS row): the
 * single-thread Math.fround restatement (oracle/js_oracle.js) on BASELINE.json
 * config 1 -- N=1,024 Plummer fixture, dt=1e-3 -- for a bounded number of steps.
 * TEST/BENCH INFRASTRUCTURE ONLY.   node oracle/js_baseline.js [steps] */
const fs = require('fs');
const path = require('path');
const oracle = require('./js_oracle.js');
const gold = path.join(__dirname, '..', 'tests', 'golden');
function loadF32(name) { const b = fs.readFileSync(path.join(gold, name + '.f32')); return new Float32Array(b.buffer, b.byteOffset, b.length / 4).slice(); }
const steps = parseInt(process.argv[2] || '40', 10);
const b0 = loadF32('plummer1024_bodies0'), v0 = loadF32('plummer1024_vel0');
oracle.runF32(b0, v0, null, 1e-3, 1.0, 2);           // warm the JIT
const t0 = process.hrtime.bigint();
oracle.runF32(b0, v0, null, 1e-3, 1.0, steps);
const secs = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({ value: 1024 * 1023 * steps / secs, unit: 'pair-interactions/s', cores: 1, kind: 'port',
  sample: 'oracle/js_oracle.js (Math.fround, single thread, node ' + process.version + '): N=1024 Plummer fixture, ' + steps + ' steps, ' + secs.toFixed(2) + ' s' }));
