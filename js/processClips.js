'use strict';
// Raw-clip counterpart of the reference's batch script (scripts/processVideos.js:110-171): a manifest
// of clips with the issues to fix, one clip after another, the same per-clip report.  The reference
// decodes and encodes through ffmpeg; here clips are raw yuv420p files and only the steps that touch
// pixels or frame counts happen: resize (`.size('720x1280')` when the clip is smaller, :96,:126) and
// trim (`targetDuration`, `trimStart`, :84-92).  Codec issues are reported and left to the host codec.
//
//   node js/processClips.js manifest.json [outDir]
//   manifest: { "clip.yuv": { width, height, fps, issues: ["resolution","duration"], targetDuration, trimStart, newName } }
const fs = require('fs');
const path = require('path');
const ffmpeg = require('./index.js');

const TARGET = { width: 720, height: 1280 };          // scripts/processVideos.js:96

// What `ffprobe -show_entries stream=codec_name,width,height,duration -of json` (:120) would say of a raw clip.
function probe(file, meta) {
  const bytes = fs.statSync(file).size;
  const frame = ffmpeg.frameBytes(meta.width, meta.height);
  if (!frame || bytes % frame) throw new Error(`INVALID_INPUT: ${file} is not a whole number of ${meta.width}x${meta.height} yuv420p frames`);
  const frames = bytes / frame;
  return { streams: [{ codec_name: 'rawvideo', width: meta.width, height: meta.height,
                       duration: (frames / (meta.fps || 30)).toFixed(6), nb_frames: frames }] };
}

function transcodeClip(inputPath, outputPath, meta, options) {
  return new Promise((resolve, reject) => {
    // the reference's own call chain (scripts/processVideos.js:71-108): trim options first, then size
    let command = ffmpeg(inputPath).inputSize(meta.width, meta.height).inputFPS(meta.fps || 30);
    if (options.duration !== undefined) command = command.duration(options.duration);   // :85-87
    if (options.trimStart) command = command.seekInput(options.trimStart);             // :90-92
    if (options.scale) command = command.size(`${TARGET.width}x${TARGET.height}`);  // :96
    command.on('end', () => resolve()).on('error', reject).save(outputPath);
  });
}

async function processClip(dir, outDir, filename, config) {
  const inputPath = path.join(dir, filename);
  const outputPath = path.join(outDir, config.newName || filename);
  const stream = probe(inputPath, config).streams[0];
  const needsResize = (config.issues || []).includes('resolution') && (stream.width < TARGET.width || stream.height < TARGET.height);
  const needsTrim = (config.issues || []).includes('duration') && config.targetDuration;
  await transcodeClip(inputPath, outputPath, config, {
    duration: needsTrim ? config.targetDuration : undefined,                        // :131
    trimStart: config.trimStart,
    scale: needsResize,
  });
  const outMeta = needsResize ? Object.assign({}, config, TARGET) : config;
  const processed = probe(outputPath, outMeta).streams[0];
  return { codec: processed.codec_name, resolution: `${processed.width}x${processed.height}`, duration: processed.duration };
}

async function main(manifestPath, outDir) {
  const manifest = JSON.parse(fs.readFileSync(manifestPath, 'utf8'));
  const dir = path.dirname(manifestPath);
  outDir = outDir || path.join(dir, 'processed');
  fs.mkdirSync(outDir, { recursive: true });
  const report = {};
  for (const [filename, config] of Object.entries(manifest)) {           // sequential, like :167-169
    try {
      report[filename] = await processClip(dir, outDir, filename, config);
    } catch (error) {
      report[filename] = { error: error.message };                        // log and carry on, like :147-149
    }
  }
  return report;
}

module.exports = { probe, processClip, main, TARGET };
if (require.main === module) {
  main(process.argv[2], process.argv[3]).then((r) => console.log(JSON.stringify(r, null, 1)),
                                              (e) => { console.error(e.message); process.exit(1); });
}
