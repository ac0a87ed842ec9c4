'use strict';
/* Throughput of the JavaScript host path (same engine, driven from Node):
 *   node tests/js/node_bench.js [N] [steps]
 * Prints one JSON line.  Synthetic uniform-cube cloud generated in JS. */
const path = require('path');
const nb = require(path.join(__dirname, '..', '..', 'nbody3d-webgpu_amd', 'js', 'nbody3d_hip.js'));
const n = parseInt(process.argv[2] || '262144', 10), steps = parseInt(process.argv[3] || '10', 10);
let seed = 12345;
function rnd() { seed = (seed + 0x6D2B79F5) | 0; let t = Math.imul(seed ^ (seed >>> 15), 1 | seed); t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t; return ((t ^ (t >>> 14)) >>> 0) / 4294967296; }
const bodies = new Float32Array(4 * n), vel = new Float32Array(4 * n);
for (let i = 0; i < n; i++) { bodies[4 * i] = 2 * rnd() - 1; bodies[4 * i + 1] = 2 * rnd() - 1; bodies[4 * i + 2] = 2 * rnd() - 1; bodies[4 * i + 3] = (0.5 + rnd()) / n; }
const sim = nb.init([bodies, vel], { G: 1.0, dt: 1e-3 });
nb.simulate(2); sim.sync();
const t0 = process.hrtime.bigint();
for (let s = 0; s < steps; s++) nb.step(1e-3);       // one FFI call per frame, as the reference's render() would
sim.sync();
const secs = Number(process.hrtime.bigint() - t0) / 1e9;
const t1 = process.hrtime.bigint();
nb.simulate(steps, 1e-3);                             // one FFI call for all frames (HIP-graph replay inside)
sim.sync();
const secsBatch = Number(process.hrtime.bigint() - t1) / 1e9;
const pairs = n * (n - 1) * steps / secs;
console.log(JSON.stringify({ host: 'node ' + process.version, n: n, steps: steps, ms_per_step: 1e3 * secs / steps, ms_per_step_simulate: 1e3 * secsBatch / steps, pairs_per_s: pairs, frac_fp32_roofline: pairs / 7.865e12, variant: sim.variant() }));
sim.destroy();
