'use strict';
// Raw-clip counterpart of the reference's loop script (scripts/loopVideos.js:42-159): extend a clip by
// repeating a middle segment.  The reference cuts three temporary files with `-ss/-t` and joins them
// with the concat demuxer (:56-100); on raw yuv420p the same result is a list of frame numbers
// (ttscale.loopOrder), so the clip is read once and written once.  The `crossfade` entry of the
// reference's strategy (:20,:31) is accepted and, as there (:55), not used.
//
//   node js/loopClips.js manifest.json [dir]
//   manifest: { "clip.yuv": { width, height, fps, loopStrategy: { extractStart, extractDuration, repeats } } }
// Each clip is replaced by its extended version (:150); the report gives the final duration (:111-113).
const fs = require('fs');
const path = require('path');
const ffmpeg = require('./index.js');

function createLoopedClip(inputPath, outputPath, meta, config) {
  const { extractStart, extractDuration, repeats } = config;
  const frame = ffmpeg.frameBytes(meta.width, meta.height);
  const src = fs.readFileSync(inputPath);
  if (!frame || src.length % frame) throw new Error(`INVALID_INPUT: ${inputPath} is not a whole number of ${meta.width}x${meta.height} yuv420p frames`);
  const [num, den] = ffmpeg.fpsRational(meta.fps || 30);
  const order = ffmpeg.loopOrder(src.length / frame, num, den, extractStart, extractDuration, repeats);
  const out = fs.openSync(outputPath, 'w');
  try {
    // consecutive source frames go out as one write
    for (let j = 0; j < order.length;) {
      let len = 1;
      while (j + len < order.length && order[j + len] === order[j] + len) len++;
      fs.writeSync(out, src, order[j] * frame, len * frame);
      j += len;
    }
  } finally {
    fs.closeSync(out);
  }
  return { frames: order.length, duration: (order.length * den / num).toFixed(6) };
}

function main(manifestPath, dir) {
  const manifest = JSON.parse(fs.readFileSync(manifestPath, 'utf8'));
  dir = dir || path.dirname(manifestPath);
  const report = {};
  for (const [filename, config] of Object.entries(manifest)) {
    const inputPath = path.join(dir, filename);
    const outputPath = path.join(dir, `extended_${filename}`);
    if (!fs.existsSync(inputPath)) { report[filename] = { error: 'Input video not found' }; continue; }     // :142-145
    try {
      report[filename] = createLoopedClip(inputPath, outputPath, config, config.loopStrategy);
      fs.renameSync(outputPath, inputPath);                                                                // :150
    } catch (error) {
      report[filename] = { error: error.message };
    }
  }
  return report;
}

module.exports = { createLoopedClip, main };
if (require.main === module) console.log(JSON.stringify(main(process.argv[2], process.argv[3]), null, 1));
