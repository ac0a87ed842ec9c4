'use strict';
/*
 * Generates tests/golden/galaxy_ref_{bodies0,vel0}.f32 by RUNNING the reference's
 * own initial-condition generator: the text of `generateGalaxy`
 * (/root/reference/nbody3d.js:51-133), `class vec3` (matrix.js) and
 * `massToRadius` / `randRange` (util.js:123,131) is read from /root/reference
 * at generation time and evaluated in a Node `vm` sandbox with stub objects for
 * the DOM globals it touches (ui, uni, camera, defaults) and a seeded
 * replacement for Math.random.  Nothing of the reference is copied into this
 * repo: only the produced arrays (data) and this script are committed.
 *
 *   node tests/golden/make_galaxy_fixture.js        (needs /root/reference)
 *   node tests/golden/make_galaxy_fixture.js full   the reference's DEFAULT system (index.html:68-74: 2 galaxies x
 *                                                   20,000 bodies -> N = 40,002): only galaxy40002_params.json is
 *                                                   written -- the SHA-256 of the two arrays, their first and last rows
 *                                                   and the galaxy list -- which pins js/ic.js::galaxies at full size
 *                                                   (tests/test_round3_cpu.py) without committing 1.3 MB of state
 *
 * The galaxy list is drawn the way main() draws it (nbody3d.js:167-175) from
 * the same seeded stream, with the UI inputs fixed below.
 */
const fs = require('fs');
const path = require('path');
const vm = require('vm');

const REF = '/root/reference';
const OUT = __dirname;
const FULL = process.argv[2] === 'full';
const PARAMS = FULL ? { seed: 40002, numGalaxies: 2, minBodies: 20000, maxBodies: 20000, outerHeight: 1080, G: 1e-4 }
                    : { seed: 20250725, numGalaxies: 2, minBodies: 380, maxBodies: 420, outerHeight: 1080, G: 1e-4 };

function mulberry32(a) {
  return function () {
    a = (a + 0x6D2B79F5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

/* text of a top-level `function name(` / `class name {` / `const name = ...;` item */
function extractBlock(src, startRe) {
  const m = startRe.exec(src);
  if (!m) throw new Error('not found: ' + startRe);
  let i = src.indexOf('{', m.index), depth = 0;
  for (; i < src.length; i++) {
    if (src[i] === '{') depth++;
    else if (src[i] === '}') { depth--; if (depth === 0) break; }
  }
  return src.slice(m.index, i + 1);
}
function extractLine(src, re) { const m = re.exec(src); if (!m) throw new Error('not found: ' + re); return m[0]; }

const nbodySrc = fs.readFileSync(path.join(REF, 'nbody3d.js'), 'utf8');
const utilSrc = fs.readFileSync(path.join(REF, 'util.js'), 'utf8');
const matrixSrc = fs.readFileSync(path.join(REF, 'matrix.js'), 'utf8');

const code = [
  extractBlock(matrixSrc, /class vec3\s*\{/),
  extractLine(utilSrc, /const massToRadius = .*;/),
  extractLine(utilSrc, /const randRange = .*;/),
  extractBlock(nbodySrc, /function generateGalaxy\s*\(/),
  // main()'s galaxy list, nbody3d.js:167-175, restated against the same stubs
  'var galaxySettings = [];',
  'for (let i = 0; i < ui.numGalaxies.value; i++) { galaxySettings.push([',
  '  [randRange(-5, 5), randRange(-5, 5), randRange(-5, 5)],',
  '  [randRange(-10, 10), randRange(-10, 10), randRange(-10, 10)],',
  '  [Math.random(), Math.random(), Math.random()],',
  '  randRange(2, 5),',
  '  Math.round(randRange(ui.minBodies.value, ui.maxBodies.value)) ]); }',
  'var result = generateGalaxy(galaxySettings);',
].join('\n');

const rng = mulberry32(PARAMS.seed);
const sandboxMath = Object.create(Math);
sandboxMath.random = rng;
const sandbox = {
  Math: sandboxMath, Float32Array: Float32Array,
  ui: { numGalaxies: { value: PARAMS.numGalaxies }, minBodies: { value: PARAMS.minBodies }, maxBodies: { value: PARAMS.maxBodies }, nBodies: {} },
  uni: { nBodies: { set: function () {} } }, camera: {}, defaults: {},
  G: PARAMS.G, nBodies: 0, sizeFactor: PARAMS.outerHeight,
};
vm.createContext(sandbox);
vm.runInContext(code, sandbox, { filename: 'reference-generateGalaxy.vm.js' });

const bodies = sandbox.result[0], vel = sandbox.result[1];
const n = bodies.length / 4;
if (n !== sandbox.nBodies) throw new Error('nBodies mismatch');
for (let i = 0; i < bodies.length; i++) if (!isFinite(bodies[i]) || !isFinite(vel[i])) throw new Error('non-finite output');
const bBuf = Buffer.from(bodies.buffer, bodies.byteOffset, bodies.byteLength), vBuf = Buffer.from(vel.buffer, vel.byteOffset, vel.byteLength);
if (!FULL) {
  fs.writeFileSync(path.join(OUT, 'galaxy_ref_bodies0.f32'), bBuf);
  fs.writeFileSync(path.join(OUT, 'galaxy_ref_vel0.f32'), vBuf);
}
const sha = function (buf) { return require('crypto').createHash('sha256').update(buf).digest('hex'); };
const meta = Object.assign({}, PARAMS, FULL ? {
  sha256_bodies0: sha(bBuf), sha256_vel0: sha(vBuf),
  first_rows: { bodies: Array.from(bodies.slice(0, 8)), vel: Array.from(vel.slice(0, 8)) },
  last_rows: { bodies: Array.from(bodies.slice(bodies.length - 8)), vel: Array.from(vel.slice(vel.length - 8)) },
} : {}, {
  n: n, source: 'text of generateGalaxy (/root/reference/nbody3d.js:51-133) evaluated under node vm; prng mulberry32',
  galaxySettings: sandbox.galaxySettings.map(function (g) { return [Array.from(g[0]), Array.from(g[1]), Array.from(g[2]), g[3], g[4]]; }),
});
fs.writeFileSync(path.join(OUT, FULL ? 'galaxy40002_params.json' : 'galaxy_ref_params.json'), JSON.stringify(meta, null, 1));
console.log(JSON.stringify({ n: n, counts: meta.galaxySettings.map(function (g) { return g[4]; }) }));
