'use strict';
/* Node-side tests, driven by tests/test_node_host.py.
 *   node tests/js/node_tests.js cpu   -> oracle + wrapper checks, no GPU
 *   node tests/js/node_tests.js gpu   -> parity of the JS host path on the GPU
 * Prints one JSON object; exit code 0 iff every check passed. */
const fs = require('fs');
const path = require('path');
const ROOT = path.join(__dirname, '..', '..');
const nb = require(path.join(ROOT, 'nbody3d-webgpu_amd', 'js', 'nbody3d_hip.js'));
const oracle = require(path.join(ROOT, 'oracle', 'js_oracle.js'));
const GOLD = path.join(ROOT, 'tests', 'golden');

function loadF32(name) { const b = fs.readFileSync(path.join(GOLD, name + '.f32')); return new Float32Array(b.buffer, b.byteOffset, b.length / 4).slice(); }
function loadF64(name) { const b = fs.readFileSync(path.join(GOLD, name + '.f64')); return new Float64Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.length)); }
function bitsEqual(a, b) { if (a.length !== b.length) return false; const x = new Uint32Array(a.buffer, a.byteOffset, a.length), y = new Uint32Array(b.buffer, b.byteOffset, b.length); for (let i = 0; i < x.length; i++) if (x[i] !== y[i]) return false; return true; }
function relPosErr(x, ref, rScale) { let m = 0; for (let i = 0; i < x.length / 4; i++) { let d = 0, r2 = 0; for (let c = 0; c < 3; c++) { d = Math.max(d, Math.abs(x[4 * i + c] - ref[4 * i + c])); r2 += ref[4 * i + c] * ref[4 * i + c]; } m = Math.max(m, d / Math.max(Math.sqrt(r2), rScale)); } return m; }

const manifest = JSON.parse(fs.readFileSync(path.join(GOLD, 'manifest.json'), 'utf8'));
const results = {}; let ok = true;
function check(name, cond, info) { results[name] = { pass: !!cond, info: info }; if (!cond) ok = false; }
function throws(fn, re) { try { fn(); } catch (e) { return re.test(String(e.message) + ' ' + String(e.code)); } return false; }

const mode = process.argv[2] || 'cpu';
const m = manifest.plummer1024;
const b0 = loadF32('plummer1024_bodies0'), v0 = loadF32('plummer1024_vel0');

if (mode === 'cpu') {
  // 1. JS oracle reproduces the C oracle's committed vectors bit for bit (10 steps)
  const t0 = Date.now();
  const r = oracle.runF32(b0, v0, null, m.dt, m.G, 10);
  const secs = (Date.now() - t0) / 1e3;
  check('js_oracle_bit_exact_vs_golden_s10', bitsEqual(r.bodies, loadF32('plummer1024_s10_bodies')) && bitsEqual(r.vel, loadF32('plummer1024_s10_vel')) && bitsEqual(r.accel, loadF32('plummer1024_s10_accel')), { seconds: secs, pairs_per_s: 10 * 1024 * 1023 / secs });
  const d = oracle.runF32(loadF32('disk771_bodies0'), loadF32('disk771_vel0'), null, manifest.disk771.dt, manifest.disk771.G, 1);
  check('js_oracle_bit_exact_disk771_s1', bitsEqual(d.bodies, loadF32('disk771_s1_bodies')) && bitsEqual(d.accel, loadF32('disk771_s1_accel')));
  // 2. wrapper loads the addon + engine library, and reports a missing GPU as an Error
  check('addon_loads', nb.load() === 2);   // NB_ABI_VERSION
  check('surface', ['init', 'step', 'simulate', 'read'].every(function (k) { return typeof nb[k] === 'function'; }) && typeof nb.Simulation === 'function');
  if (nb.deviceCount() === 0) {
    check('no_device_throws', throws(function () { nb.init([b0, v0]); }, /no HIP device.*NB_2|NB_2/));
  }
  // 2a. the launch planner from JavaScript, no GPU needed (nb_plan_query)
  const pq = nb.planQuery({ n: 262144, nCU: 256, clockHz: 2.4e9 }), pq2 = nb.planQuery({ n: 1024, nCU: 256, clockHz: 2.4e9 });
  check('plan_query', /^f32pk_symw_ipl16_j1_w2048/.test(pq.variant) && pq.sym === 1 && pq.symRows === 262144 && pq.layerBytes === 12 * 262144 * pq.symLayers &&
    /^f32pk_fused_regs1024/.test(pq2.variant) && pq2.sym === 0 && throws(function () { nb.planQuery({ n: 0, nCU: 256, clockHz: 2.4e9 }); }, /NB_1/), pq);
  check('step_before_init_throws', throws(function () { new nb.Simulation().step(1e-3); }, /not initialised/));
  check('bad_particles_throws', throws(function () { new nb.Simulation().init({}); }, /expected/));
  // 2b. the seedable generateGalaxy port reproduces, bit for bit, the arrays the
  //     reference's own generator text produced from the same random stream
  const ic = require(path.join(ROOT, 'nbody3d-webgpu_amd', 'js', 'ic.js'));
  const gp = JSON.parse(fs.readFileSync(path.join(GOLD, 'galaxy_ref_params.json'), 'utf8'));
  const stream = ic.mulberry32(gp.seed);
  const list = ic.galaxySettings(gp.numGalaxies, { random: stream, minBodies: gp.minBodies, maxBodies: gp.maxBodies });
  const gal = ic.galaxies(list, { random: stream, G: gp.G, sizeFactor: gp.outerHeight });
  check('galaxy_port_settings', JSON.stringify(list) === JSON.stringify(gp.galaxySettings));
  check('galaxy_port_bit_exact_vs_reference_generator', gal[0].length === 4 * gp.n && bitsEqual(gal[0], loadF32('galaxy_ref_bodies0')) && bitsEqual(gal[1], loadF32('galaxy_ref_vel0')));
  const pl = ic.plummer(2048, { seed: 3 }); let msum = 0, cx = 0;
  for (let i = 0; i < 2048; i++) { msum += pl[0][4 * i + 3]; cx += pl[0][4 * i]; }
  check('plummer_js_sane', Math.abs(msum - 1) < 1e-5 && Math.abs(cx / 2048) < 1e-6 && pl[1][3] === 0);
  // 3. pause semantics (util.js:36-64)
  const s = new nb.Simulation({ dt: 1e-3 });
  s.togglePause(); const paused = s.dt === 0; s.setDt(2e-3); const still = s.dt === 0; s.togglePause();
  check('pause_semantics', paused && still && s.dt === 2e-3);
} else {
  // GPU: the JS host path end to end on the golden fixture
  const sim = nb.init([b0, v0], { G: m.G, dt: m.dt });
  nb.simulate(10);
  const s10 = nb.read();
  check('gpu_s10_vs_oracle', relPosErr(s10.bodies, loadF32('plummer1024_s10_bodies'), m.r_scale) < 1e-6);
  for (let k = 0; k < 90; k++) nb.step(m.dt);
  const s100 = nb.read();
  const e64 = relPosErr(s100.bodies, loadF64('plummer1024_s100_bodies'), m.r_scale);
  check('gpu_s100_vs_f64_oracle_le_1e-4', e64 < 2e-5, { err: e64, variant: sim.variant() });
  // dt = 0 / pause is a no-op (nbody3d.js:474)
  sim.togglePause(); nb.step(); const p = nb.read(); sim.togglePause();
  check('pause_noop', bitsEqual(p.bodies, s100.bodies) && bitsEqual(p.vel, s100.vel) && bitsEqual(p.accel, s100.accel));
  // export / import round trip (util.js:186-263)
  const json = sim.exportJSON();
  const sim2 = new nb.Simulation({ dt: m.dt }).importJSON(json);
  check('json_roundtrip_G', Math.abs(sim2.G - 1.0) < 1e-12 && sim2.nBodies === 1024);
  sim.step(); sim2.step();
  check('json_roundtrip_continues_identically', bitsEqual(sim.read().bodies, sim2.read().bodies));
  // a checkpoint in the reference's full schema (camera block, plain arrays, G as a
  // log10 string, util.js:186-201) restores and steps; the camera block is ignored
  const refJson = JSON.parse(json);
  refJson.camera = { target: [0, 0, 0], position: [0, 0, 10], radius: 10, azimuth: 0, elevation: 0, fov: 60, near: 0.1, far: 1e5 };
  refJson.G = '-4.00';
  const sim3 = new nb.Simulation({ dt: 1e-4 }).importJSON(JSON.stringify(refJson));
  check('reference_schema_import', Math.abs(sim3.G - 1e-4) < 1e-18 && sim3.nBodies === 1024);
  sim3.step(); const after = sim3.read();
  check('reference_schema_steps', isFinite(after.bodies[0]) && after.accel[3] === 0);
  // ... and the camera block comes back out untouched (browser -> engine -> browser keeps the view, util.js:190-199,246-256),
  // in the reference's key order (bodies, vel, accel, camera, G); a state that never had one exports none
  const back = JSON.parse(sim3.exportJSON());
  check('camera_block_passes_through', JSON.stringify(back.camera) === JSON.stringify(refJson.camera) && back.G === '-4.00' &&
    JSON.stringify(Object.keys(back)) === JSON.stringify(['bodies', 'vel', 'accel', 'camera', 'G']), Object.keys(back));
  check('no_camera_no_block', JSON.parse(json).camera === undefined);
  sim3.destroy();
  // wrong array length is an error, not a crash
  check('bad_length_throws', throws(function () { sim.restore({ bodies: new Float32Array(8), vel: new Float32Array(8) }); }, /expected/));
  const d = sim.diagnostics();
  check('diagnostics', isFinite(d.kinetic) && d.potential < 0 && d.momentum.length === 3, d);
  // single-process multi-device handle from JS: 4 i-shards (virtual shards when fewer GPUs)
  const sm = new nb.Simulation({ G: m.G, dt: m.dt, shards: 4 }).init([b0, v0]);
  sm.simulate(100);
  const em = relPosErr(sm.read().bodies, loadF64('plummer1024_s100_bodies'), m.r_scale);
  check('gpu_multi_shard_handle_vs_f64_oracle', em < 2e-5, { err: em, variant: sm.variant() });
  const dm = sm.diagnostics();
  check('multi_diagnostics_match_single', Math.abs(dm.kinetic - d.kinetic) < 1e-3 * Math.abs(d.kinetic) && dm.potential < 0, dm);
  check('multi_has_no_kernel_timing', throws(function () { sm.kernelTimes(); }, /not available/));
  sm.destroy();
  // the same handle in RCCL mode (ncclCommInitAll + grouped in-place ncclAllGather, SURVEY.md §8(e)):
  // one shard on the one GPU of a test box -- RCCL refuses two ranks on one device
  const sr = new nb.Simulation({ G: m.G, dt: m.dt, shards: 1, collective: 'rccl' }).init([b0, v0]);
  const ci = sr.collectiveInfo();
  check('multi_rccl_mode_info', ci.mode === 'rccl' && ci.nranks === 1 && ci.rcclVersion > 20000, ci);
  sr.simulate(100);
  const er = relPosErr(sr.read().bodies, loadF64('plummer1024_s100_bodies'), m.r_scale);
  check('gpu_multi_rccl_handle_vs_f64_oracle', er < 2e-5, { err: er, variant: sr.variant() });
  sr.destroy();
  check('multi_rccl_refuses_shared_device', nb.deviceCount() >= 2 ||
    throws(function () { new nb.Simulation({ shards: 2, collective: 'rccl' }).init([b0, v0]); }, /own device/));
  // viewer frame feed (nbody3d.js:408-415,482-487; colour input :380): snapshot == read() of that step,
  // later steps do not disturb it
  const sf = new nb.Simulation({ G: m.G, dt: m.dt }).init([b0, v0]);
  sf.simulate(7);
  const at7 = sf.read();
  sf.requestFrame();
  sf.simulate(5);
  const fr = sf.frame(true);
  let speedOk = fr !== null && fr.step === 7 && bitsEqual(fr.bodies, at7.bodies);
  for (let i = 0; speedOk && i < 1024; i += 37) {
    const vx = at7.vel[4 * i], vy = at7.vel[4 * i + 1], vz = at7.vel[4 * i + 2];
    const want = Math.sqrt(Math.fround(Math.fround(Math.fround(vx * vx) + Math.fround(vy * vy)) + Math.fround(vz * vz)));
    speedOk = Math.abs(fr.speed[i] - want) <= 4e-7 * Math.max(want, 1e-30);
  }
  check('frame_feed_equals_read', speedOk, fr ? { step: fr.step } : null);
  check('frame_before_request_throws', throws(function () { new nb.Simulation({}).init([b0, v0]).frame(false); }, /nb_frame_request has not been called/));
  const st = sf.enableTiming(true).simulate(3).stepTimes();
  check('step_times', st.launches === 3 && st.forceMs > 0 && st.exchangeMs === 0, st);
  sf.destroy();
  // f64 simulation takes Float64Array
  const s64 = new nb.Simulation({ f64: true, G: m.G, dt: m.dt }).init([Float64Array.from(b0), Float64Array.from(v0)]);
  s64.simulate(100);
  const e = relPosErr(s64.read().bodies, loadF64('plummer1024_s100_bodies'), m.r_scale);
  check('gpu_f64_vs_f64_oracle', e < 1e-12, { err: e });
  s64.destroy(); sim2.destroy(); sim.destroy();
}
console.log(JSON.stringify({ ok: ok, mode: mode, results: results }));
process.exit(ok ? 0 : 1);
