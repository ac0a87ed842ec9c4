'use strict';
/* The reference's DEFAULT workload driven from Node: two galaxies of 20,000 bodies
 * + central masses = N = 40,002 (index.html:68-74; nbody3d.js:60), G = dt = 1e-4
 * (nbody3d.js:6-7), one step() per frame as render() does.  Prints one JSON line. */
const path = require('path');
const root = path.join(__dirname, '..', '..', 'nbody3d-webgpu_amd', 'js');
const nb = require(path.join(root, 'nbody3d_hip.js'));
const ic = require(path.join(root, 'ic.js'));
const frames = parseInt(process.argv[2] || '500', 10);
const feed = process.argv[3] === 'feed' || process.argv[3] === 'feednoacq';
const noacq = process.argv[3] === 'feednoacq';   // also take a viewer snapshot every frame (requestFrame + frame(false))
const rand = ic.mulberry32(7);
const particles = ic.galaxies(ic.galaxySettings(2, { random: rand, minBodies: 20000, maxBodies: 20000 }), { random: rand });
const n = particles[0].length / 4;
const sim = nb.init(particles);                      // defaults G = dt = 1e-4
const d0 = sim.diagnostics();
nb.simulate(20); sim.sync();
const t0 = process.hrtime.bigint();
let landed = 0;
for (let f = 0; f < frames; f++) {
  nb.step();
  if (feed) { sim.requestFrame(); if (!noacq && sim.frame(false)) landed++; }
}
sim.sync();
const secs = Number(process.hrtime.bigint() - t0) / 1e9;
const d1 = sim.diagnostics();
const out = nb.read();
let finite = true;
for (let i = 0; i < out.bodies.length; i++) if (!isFinite(out.bodies[i])) { finite = false; break; }
console.log(JSON.stringify({
  workload: 'reference default: 2 galaxies x 20000 + 2 central masses', n: n, frames: frames,
  ms_per_frame: 1e3 * secs / frames, frames_per_s: frames / secs, pairs_per_s: n * (n - 1) * frames / secs,
  frac_fp32_roofline: n * (n - 1) * frames / secs / 7.865e12, variant: sim.variant(), finite: finite,
  E0: d0.kinetic + d0.potential, E1: d1.kinetic + d1.potential, frame_feed: feed, frames_landed_in_loop: landed,
}));
sim.destroy();
